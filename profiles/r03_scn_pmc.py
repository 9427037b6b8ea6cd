#!/usr/bin/env python3
"""A few 20-step rollout launches of one landmark scenario, for counter passes around it:
   rocprofv3 --pmc ... -- python3 profiles/r03_scn_pmc.py <scenario> <N> <B>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402

scenario, N, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
K = 20
env = formation_gym.make_env(scenario, False, N, num_envs=B, device="cuda:0")
env.seed(1); env.reset(); env.auto_reset = True
acts = (torch.rand((K, B, N, 2), device="cuda") * 2 - 1).contiguous()
D = env._out["obs"].shape[-1]
f = dict(dtype=torch.float32, device="cuda")
out = dict(obs=torch.empty((K, B, N, D), **f), reward=torch.empty((K, B, N), **f), indiv=torch.empty((K, B, N), **f),
           done=torch.zeros((K, B, N), dtype=torch.uint8, device="cuda"))
for _ in range(6):
    env.rollout(acts, out=out)
torch.cuda.synchronize()
