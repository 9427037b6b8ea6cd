#!/usr/bin/env python3
"""[study script: written against the study-time arena API (fg_arena_create with an initial mapping, fg_arena_view,
fg_arena_keep; library of commits 77e8adc ... f312e76) - the shipped API maps one candidate at a time, include/formation_hip.h]
Do arbitrary COMBINATIONS of chunks beat the best contiguous window?  Arena of 1 GiB chunks; contiguous windows
first, then views (fg_arena_view) of: the best window reversed, random chunk selections, and a greedy search that swaps
single chunks of the best window for unused ones.   python profiles/r03_place/scan_combos.py N B K arena_GB"""
import ctypes
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402
from formation_gym import _native                         # noqa: E402

N, B, K, GB = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
dev = "cuda:0"
env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
env.scenario.reset_device(env.world, rng_offset=3)
env.auto_reset = True
acts = torch.zeros((K, B, N, 2), device=dev)
small = dict(reward=torch.empty((K, B, N), device=dev), indiv=torch.empty((K, B, N), device=dev),
             done=torch.zeros((K, B, N), dtype=torch.uint8, device=dev))
nfl = K * B * N * 6 * N
bytes_launch = (24 * N * N + 53 * N + 16) * B * K
lib = _native.load()
lib.fg_arena_view.restype = ctypes.c_int
lib.fg_arena_view.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]


class Raw(object):
    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (nfloats,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def time_ptr(ptr, reps=3):
    obs = torch.as_tensor(Raw(ptr, nfl), device=dev).view(K, B, N, 6 * N)
    out = dict(small, obs=obs)
    env.rollout(acts, out=out)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for r in range(reps):
        env.rollout(acts, out=out)
        e[r + 1].record()
    torch.cuda.synchronize()
    env._roll_launchers.clear()
    ms = sorted(e[r].elapsed_time(e[r + 1]) for r in range(reps))
    return ms[len(ms) // 2]


arena, base, chunk = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
_native.check(lib.fg_arena_create(0, int(GB * 1e9), 1 << 30, ctypes.byref(arena), ctypes.byref(base), ctypes.byref(chunk)))
chunk = chunk.value
n = -(-int(GB * 1e9) // chunk)
W = -(-nfl * 4 // chunk)
gbs = lambda ms: bytes_launch / (ms * 1e-3) / 1e9


def view(idx):
    arr = (ctypes.c_uint32 * len(idx))(*idx)
    p = ctypes.c_void_p()
    _native.check(lib.fg_arena_view(arena, arr, len(idx), ctypes.byref(p)))
    return p.value


win = [(time_ptr(base.value + k * chunk), k) for k in range(n - W + 1)]
ms_best, k_best = min(win)
srt = sorted(m for m, _ in win)
print("N=%d B=%d K=%d window %d chunks of 1 GiB, %d chunks: contiguous windows best %.0f median %.0f worst %.0f GB/s (best at chunk %d)" % (
    N, B, K, W, n, gbs(srt[0]), gbs(srt[len(srt) // 2]), gbs(srt[-1]), k_best))
best = list(range(k_best, k_best + W))
print("best window through a view (same order): %.0f GB/s;  reversed: %.0f GB/s" % (gbs(time_ptr(view(best))), gbs(time_ptr(view(best[::-1])))))
rnd = random.Random(1)
r = sorted(gbs(time_ptr(view(rnd.sample(range(n), W)))) for _ in range(24))
print("24 random selections of %d chunks: min %.0f median %.0f max %.0f GB/s" % (W, r[0], r[len(r) // 2], r[-1]))
r = sorted(gbs(time_ptr(view(sorted(rnd.sample(range(n), W))))) for _ in range(24))
print("24 random selections, ascending chunk order: min %.0f median %.0f max %.0f GB/s" % (r[0], r[len(r) // 2], r[-1]))
# greedy: replace one chunk of the current best combination by an unused chunk whenever that is faster
cur, cur_ms = best[:], time_ptr(view(best))
for sweep in range(2):
    improved = False
    for j in range(W):
        cand = [c for c in range(n) if c not in cur]
        rnd.shuffle(cand)
        for c in cand[:10]:
            trial = cur[:]; trial[j] = c
            ms = time_ptr(view(trial))
            if ms < cur_ms * 0.995:
                cur, cur_ms, improved = trial, ms, True
    print("greedy sweep %d: %.0f GB/s with chunks %s" % (sweep, gbs(cur_ms), cur))
    if not improved:
        break
print("re-timed: best contiguous %.0f GB/s, greedy result %.0f GB/s" % (gbs(time_ptr(base.value + k_best * chunk, 7)), gbs(time_ptr(view(cur), 7))))
torch.cuda.synchronize()
_native.check(lib.fg_arena_destroy(arena))
