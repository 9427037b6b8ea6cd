#!/usr/bin/env python3
"""Small fixed workload for PMC passes: diag.py {step N B | fill MB} - 30 launches."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gym-formation_amd"))
kind = sys.argv[1]
if kind == "fill":
    x = torch.empty(int(sys.argv[2]) * (1 << 20) // 4, device="cuda")
    for _ in range(30):
        x.fill_(1.0)
    torch.cuda.synchronize()
else:
    import formation_gym
    N, B = int(sys.argv[2]), int(sys.argv[3])
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
    env.seed(1); env.reset()
    act = (torch.rand((B, N, 2), device="cuda") * 2 - 1).contiguous()
    launch = env.scenario.bind_step(env.world, act, env._out, auto_reset=False)
    for _ in range(30):
        launch(0)
    torch.cuda.synchronize()
