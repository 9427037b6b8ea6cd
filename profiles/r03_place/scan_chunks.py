#!/usr/bin/env python3
"""[study script: written against the study-time arena API (fg_arena_create with an initial mapping, fg_arena_view,
fg_arena_keep; library of commits 77e8adc ... f312e76) - the shipped API maps one candidate at a time, include/formation_hip.h]
Per-chunk rates inside an arena: (a) the rollout launch confined to ONE chunk (K small), (b) torch's fill_ of the chunk.
Do they rank the chunks alike?   python profiles/r03_place_scan4.py N B Kprobe arena_GB chunk_MB"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import numpy as np                                        # noqa: E402
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402
from formation_gym import _native                         # noqa: E402

N, B, K, GB, CHUNK_MB = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
dev = "cuda:0"
env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
env.scenario.reset_device(env.world, rng_offset=3)
env.auto_reset = True
acts = torch.zeros((K, B, N, 2), device=dev)
small = dict(reward=torch.empty((K, B, N), device=dev), indiv=torch.empty((K, B, N), device=dev),
             done=torch.zeros((K, B, N), dtype=torch.uint8, device=dev))
nfl = K * B * N * 6 * N
bytes_launch = (24 * N * N + 53 * N + 16) * B * K


class Raw(object):
    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (nfloats,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def timed(fn, reps=5):
    fn()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for r in range(reps):
        fn()
        e[r + 1].record()
    torch.cuda.synchronize()
    ms = sorted(e[r].elapsed_time(e[r + 1]) for r in range(reps))
    return ms[len(ms) // 2]


lib = _native.load()
arena, base, chunk = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
_native.check(lib.fg_arena_create(0, int(GB * 1e9), CHUNK_MB << 20, ctypes.byref(arena), ctypes.byref(base), ctypes.byref(chunk)))
chunk = chunk.value
nchunks = -(-int(GB * 1e9) // chunk)
assert nfl * 4 <= chunk, "probe launch (%d MB) must fit a chunk" % (nfl * 4 >> 20)
print("N=%d B=%d K=%d probe %.0f MB in chunks of %d MiB; %d chunks" % (N, B, K, nfl * 4 / 1e6, chunk >> 20, nchunks))
whole = torch.as_tensor(Raw(base.value, nchunks * chunk // 4), device=dev)
a, f = [], []
for k in range(nchunks):
    off = k * chunk // 4
    obs = whole[off:off + nfl].view(K, B, N, 6 * N)
    out = dict(small, obs=obs)
    a.append(bytes_launch / (timed(lambda: env.rollout(acts, out=out)) * 1e-3) / 1e9)
    env._roll_launchers.clear()
    c = whole[off:off + chunk // 4]
    f.append(chunk / (timed(lambda: c.fill_(1.0)) * 1e-3) / 1e9)
a, f = np.array(a), np.array(f)
for k in range(nchunks):
    print("  chunk %3d  rollout %.0f GB/s   fill_ %.0f GB/s" % (k, a[k], f[k]))
print("rollout: min %.0f median %.0f max %.0f;  fill_: min %.0f median %.0f max %.0f;  correlation %.3f" % (
    a.min(), np.median(a), a.max(), f.min(), np.median(f), f.max(), np.corrcoef(a, f)[0, 1]))
srt = np.sort(a)[::-1]
print("mean of the fastest quarter %.0f, half %.0f, all %.0f GB/s" % (srt[:len(srt) // 4].mean(), srt[:len(srt) // 2].mean(), srt.mean()))
del whole, obs, c, out
torch.cuda.synchronize()
_native.check(lib.fg_arena_destroy(arena))
