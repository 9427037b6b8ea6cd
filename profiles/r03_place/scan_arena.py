#!/usr/bin/env python3
"""[study script: written against the study-time arena API (fg_arena_create with an initial mapping, fg_arena_view,
fg_arena_keep; library of commits 77e8adc ... f312e76) - the shipped API maps one candidate at a time, include/formation_hip.h]
Arena of separately created physical chunks (fg_arena_*, HIP virtual memory management): scan windows with the
rollout launch, keep the best window's chunks, release the others, time again.
   python profiles/r03_place_scan3.py N B K arena_GB chunk_MB"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
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


def time_buffer(obs, reps=3):
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


lib = _native.load()
arena, base, chunk = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
_native.check(lib.fg_arena_create(0, int(GB * 1e9), CHUNK_MB << 20, ctypes.byref(arena), ctypes.byref(base), ctypes.byref(chunk)))
chunk = chunk.value
nchunks = -(-int(GB * 1e9) // chunk)
W = -(-nfl * 4 // chunk)
print("N=%d B=%d K=%d: window %.2f GB = %d chunks of %d MiB; arena %d chunks at %#x" % (N, B, K, nfl * 4 / 1e9, W, chunk >> 20, nchunks, base.value))
whole = torch.as_tensor(Raw(base.value, nchunks * chunk // 4), device=dev)
whole[:1024].zero_()
res = []
step = max(1, W // 4)
for k in range(0, nchunks - W + 1, step):
    off = k * chunk // 4
    ms = time_buffer(whole[off:off + nfl].view(K, B, N, 6 * N))
    res.append((ms, k))
    print("  window at chunk %4d (%6.2f GB)  %.4f ms  %.0f GB/s" % (k, k * chunk / 1e9, ms, bytes_launch / (ms * 1e-3) / 1e9), flush=True)
ms, k = min(res)
print("best window: chunk %d, %.4f ms; worst %.4f ms" % (k, ms, max(res)[0]))
torch.cuda.synchronize()
del whole
_native.check(lib.fg_arena_keep(arena, k * chunk, nfl * 4))
free, total = torch.cuda.mem_get_info()
print("after fg_arena_keep: %.1f GB free of %.1f" % (free / 1e9, total / 1e9))
kept = torch.as_tensor(Raw(base.value + k * chunk, nfl), device=dev).view(K, B, N, 6 * N)
for _ in range(3):
    ms2 = time_buffer(kept)
    print("  kept window again: %.4f ms  %.0f GB/s" % (ms2, bytes_launch / (ms2 * 1e-3) / 1e9))
# correctness of the mapping: the launch's output through the arena equals a plain tensor's
ref = torch.empty((K, B, N, 6 * N), device=dev)
snap = env._snapshot()
env.rollout(acts, out=dict(small, obs=kept)); torch.cuda.synchronize()
env._restore(snap)
env.rollout(acts, out=dict(small, obs=ref)); torch.cuda.synchronize()
print("outputs equal:", bool(torch.equal(ref, kept)))
del kept
_native.check(lib.fg_arena_destroy(arena))
print("destroyed; free %.1f GB" % (torch.cuda.mem_get_info()[0] / 1e9))
