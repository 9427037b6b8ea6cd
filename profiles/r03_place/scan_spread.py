#!/usr/bin/env python3
"""[study script: written against the study-time arena API (fg_arena_create with an initial mapping, fg_arena_view,
fg_arena_keep; library of commits 77e8adc ... f312e76) - the shipped API maps one candidate at a time, include/formation_hip.h]
Chunks spread over a LARGE arena: random selections, stratified (one chunk per stratum) ascending / shuffled, for
several chunk sizes.   python profiles/r03_place/scan_spread.py N B K arena_GB "chunk_MiB ..." """
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


def time_ptr(ptr, reps=4):
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


def stats(v):
    v = sorted(v)
    return "min %.0f median %.0f max %.0f" % (v[0], v[len(v) // 2], v[-1])


print("N=%d B=%d K=%d buffer %.2f GB, arena %.0f GB" % (N, B, K, nfl * 4 / 1e9, GB))
rnd = random.Random(11)
for cm in CHUNKS:
    arena, base, chunk = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
    _native.check(lib.fg_arena_create(0, int(GB * 1e9), cm << 20, ctypes.byref(arena), ctypes.byref(base), ctypes.byref(chunk)))
    chunk = chunk.value
    n = -(-int(GB * 1e9) // chunk)
    W = -(-nfl * 4 // chunk)

    def view(idx):
        arr = (ctypes.c_uint32 * len(idx))(*idx)
        p = ctypes.c_void_p()
        _native.check(lib.fg_arena_view(arena, arr, len(idx), ctypes.byref(p)))
        return p.value

    contig = [gbs(time_ptr(base.value + k * chunk)) for k in range(0, n - W + 1, max(1, (n - W) // 12))]
    rand = [gbs(time_ptr(view(rnd.sample(range(n), W)))) for _ in range(12)]
    strat_asc, strat_shuf = [], []
    for _ in range(6):
        idx = [min(n - 1, int((j + rnd.random()) * n / W)) for j in range(W)]
        idx = sorted(set(idx))
        while len(idx) < W:
            c = rnd.randrange(n)
            if c not in idx:
                idx.append(c)
        idx = sorted(idx)
        strat_asc.append(gbs(time_ptr(view(idx))))
        sh = idx[:]; rnd.shuffle(sh)
        strat_shuf.append(gbs(time_ptr(view(sh))))
    print("  chunks of %4d MiB (%4d of %5d): contiguous windows %s | random %s | stratified ascending %s | stratified shuffled %s GB/s" % (
        chunk >> 20, W, n, stats(contig), stats(rand), stats(strat_asc), stats(strat_shuf)), flush=True)
    torch.cuda.synchronize()
    _native.check(lib.fg_arena_destroy(arena))
