#!/usr/bin/env python3
"""[study script: written against the study-time arena API (fg_arena_create with an initial mapping, fg_arena_view,
fg_arena_keep; library of commits 77e8adc ... f312e76) - the shipped API maps one candidate at a time, include/formation_hip.h]
The buffer as a SHUFFLED set of physical chunks: arena of exactly the buffer's size (+ slack factor), chunks of C MiB,
mapped (a) in allocation order, (b) in random order (fg_arena_view), for several chunk sizes.
   python profiles/r03_place/scan_shuffle.py N B K factor "chunk_MiB ..." """
import ctypes
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402
from formation_gym import _native                         # noqa: E402

N, B, K, FACTOR = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
CHUNKS = [int(x) for x in sys.argv[5].split()]
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
gbs = lambda ms: bytes_launch / (ms * 1e-3) / 1e9


class Raw(object):
    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (nfloats,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def time_ptr(ptr, reps=5):
    obs = torch.as_tensor(Raw(ptr, nfl), device=dev).view(K, B, N, 6 * N)
    out = dict(small, obs=obs)
    for _ in range(2):
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


print("N=%d B=%d K=%d buffer %.2f GB, arena factor %.1f" % (N, B, K, nfl * 4 / 1e9, FACTOR))
plain = torch.empty(nfl, dtype=torch.float32, device=dev)
print("  torch.empty (hipMalloc)                      %.0f GB/s" % gbs(time_ptr(plain.data_ptr())))
del plain
torch.cuda.empty_cache()
rnd = random.Random(7)
for cm in CHUNKS:
    arena, base, chunk = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
    _native.check(lib.fg_arena_create(0, int(nfl * 4 * FACTOR), cm << 20, ctypes.byref(arena), ctypes.byref(base), ctypes.byref(chunk)))
    chunk = chunk.value
    n = -(-int(nfl * 4 * FACTOR) // chunk)
    W = -(-nfl * 4 // chunk)
    res = ["order as created %.0f" % gbs(time_ptr(base.value))]
    for trial in range(3):
        idx = rnd.sample(range(n), W)
        arr = (ctypes.c_uint32 * W)(*idx)
        p = ctypes.c_void_p()
        _native.check(lib.fg_arena_view(arena, arr, W, ctypes.byref(p)))
        res.append("shuffled %.0f" % gbs(time_ptr(p.value)))
    # interleave: even chunks first half / odd second half (a deterministic de-correlation)
    idx = list(range(0, W, 2)) + list(range(1, W, 2))
    arr = (ctypes.c_uint32 * W)(*idx)
    p = ctypes.c_void_p()
    _native.check(lib.fg_arena_view(arena, arr, W, ctypes.byref(p)))
    res.append("even-then-odd %.0f" % gbs(time_ptr(p.value)))
    print("  chunks of %5d MiB (%5d of %5d): %s GB/s" % (chunk >> 20, W, n, "  ".join(res)), flush=True)
    torch.cuda.synchronize()
    _native.check(lib.fg_arena_destroy(arena))
