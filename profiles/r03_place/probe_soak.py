#!/usr/bin/env python3
"""Soak of the placement path: the same env places a fresh rollout buffer again and again (arena created, candidates
mapped / timed / unmapped, winner kept, previous arena destroyed) and a rollout into every placed buffer is compared with
the same rollout into an ordinary tensor, bit for bit; free memory and the retired address space are reported.
   python3 profiles/r03_place/probe_soak.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402
from formation_gym import _native                         # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = "cuda:0"
N, B, K = 27, 4096, 20
env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
ref = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
for e in (env, ref):
    e.seed(2); e.scenario.reset_device(e.world, rng_offset=5); e.auto_reset = True
gen = torch.Generator(device=dev); gen.manual_seed(0)
bad = 0
for r in range(rounds):
    acts = (torch.rand((K, B, N, 2), generator=gen, device=dev) * 2 - 1).contiguous()
    out = env.alloc_rollout_buffers(K)
    for a in env._arenas[:-1]:                            # the arenas of earlier rounds: their buffers are dropped here
        a.close()
    del env._arenas[:-1]
    o1, r1, d1, _ = env.rollout(acts, out=out)
    o2, r2, d2, _ = ref.rollout(acts)
    ok = torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    bad += int(not ok)
    free = torch.cuda.mem_get_info()[0] / 1e9
    p = env.placement
    print("round %2d  %s  kept '%s' %.4f ms (as created %.4f)  free %.1f GB  retired address space %.1f TB" % (
        r, "ok " if ok else "MISMATCH", p["kept"], p["kept_ms"], p["as_created_ms"], free,
        _native.load().fg_arena_retired_address_bytes() / 1e12), flush=True)
    del out, o1, o2
print("%d of %d rounds differ" % (bad, rounds))
