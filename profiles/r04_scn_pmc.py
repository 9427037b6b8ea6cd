#!/usr/bin/env python3
"""A few K-step rollout launches of one landmark scenario into an HBM-size buffer, for counter passes around it:
   rocprofv3 --pmc ... -- python3 profiles/r04_scn_pmc.py <scenario> <N> <B> [variant]     (variant 1 = the run-time-count kernel)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402

scenario, N, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
env = formation_gym.make_env(scenario, False, N, num_envs=B, device="cuda:0")
env.seed(1); env.scenario.reset_device(env.world, rng_offset=9); env.auto_reset = True
env.scenario.kernel_variant = variant
D = env._out["obs"].shape[-1]
K = 20
while K * B * N * D * 4 < 1.1e9:
    K += 20
acts = (torch.rand((K, B, N, 2), device="cuda") * 2 - 1).contiguous()
f = dict(dtype=torch.float32, device="cuda")
out = dict(obs=torch.empty((K, B, N, D), **f), reward=torch.empty((K, B, N), **f), indiv=torch.empty((K, B, N), **f),
           done=torch.zeros((K, B, N), dtype=torch.uint8, device="cuda"))
for _ in range(4):
    env.rollout(acts, out=out)
torch.cuda.synchronize()
