#!/usr/bin/env python3
"""Does the KIND of allocation change what the rollout launch sees?  The 27 x 4096 x 20 observation buffer (1.43 GB) as an
ordinary allocation, and as hipExtMallocWithFlags memory: default, fine-grained, uncached (MTYPE_UC: no write-back lines in L2),
physically contiguous.  Same launch (the 8-writer-wave instantiation of a placed buffer) into each, us/step; a dense `fill_`
beside it.   python3 profiles/r05_alloc_flags.py [N B K]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402
from formation_gym import placement                       # noqa: E402

N, B, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (27, 4096, 20)
dev = torch.device("cuda:0")
env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
env.scenario.reset_device(env.world, rng_offset=1)
env.auto_reset = True
acts = (torch.rand((K, B, N, 2), device=dev) * 2 - 1).contiguous()
f = dict(dtype=torch.float32, device=dev)
small = dict(reward=torch.empty((K, B, N), **f), indiv=torch.empty((K, B, N), **f), done=torch.zeros((K, B, N), dtype=torch.uint8, device=dev))
nfl = K * B * N * 6 * N
stream = torch.cuda.current_stream(dev)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipFree.argtypes = [ctypes.c_void_p]
flagged = [True]
placement.is_placed = lambda address: flagged[0]


def time_fn(flat):
    env.rollout(acts, out=dict(small, obs=flat.view(K, B, N, 6 * N)))
    env._roll_launchers.clear()


def fill_rate(flat):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    flat.fill_(1.0)
    e0.record()
    for _ in range(4):
        flat.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    return flat.numel() * 4 * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9


held = []
for rnd in range(2):
    for label, flags in (("torch.empty (ordinary)", None), ("hipExtMalloc default", 0x0), ("hipExtMalloc fine-grained", 0x1),
                         ("hipExtMalloc uncached", 0x3), ("hipExtMalloc contiguous", 0x4)):
        if flags is None:
            flat = torch.empty(nfl, **f)
            ptr = None
        else:
            p = ctypes.c_void_p()
            rc = hip.hipExtMallocWithFlags(ctypes.byref(p), nfl * 4, flags)
            if rc != 0 or not p.value:
                print("%-28s allocation failed (%d)" % (label, rc), flush=True)
                continue
            ptr = p.value
            flat = torch.as_tensor(placement._Raw(ptr, nfl), device=dev)
        held.append((flat, ptr))                                     # held: a freed allocation's pages would come straight back
        res = []
        for fl in (True, False):
            flagged[0] = fl
            res.append(placement._time_launch(time_fn, flat, stream, 12) * 1e3 / K)
        print("round %d  %-28s 8 writer waves %.2f us/step   4 writer waves %.2f us/step   fill_ %.0f GB/s"
              % (rnd, label, res[0], res[1], fill_rate(flat)), flush=True)
torch.cuda.synchronize()
for flat, ptr in held:
    del flat
