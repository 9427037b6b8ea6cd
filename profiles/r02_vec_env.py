#!/usr/bin/env python3
"""FormationVecEnv.step (the SubprocVecEnv / DummyVecEnv replacement, train/maddpg-v2/utils/env_wrappers.py) in its three
reset modes: us per vec-env step, episodes of 100 steps restarting all the time (HIP events around 300 steps)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402
from formation_gym.vec_env import FormationVecEnv         # noqa: E402

dev = "cuda:0"
print("| shape | reset mode | us per vec-env step | env-steps/s |")
print("|---|---|---|---|")
for N, B in ((9, 4096), (27, 4096), (27, 256), (81, 2048)):
    for mode in ("device", "device_mt", "host"):
        env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
        env.seed(1)
        venv = FormationVecEnv(env, reset_mode=mode)
        venv.reset()
        env.world.step_count.copy_((torch.arange(B, device=dev) % 100).int())     # episodes end at different steps
        if mode == "device_mt":
            venv.ts[:] = env.world.step_count.cpu().numpy()
        act = torch.rand((B, N, 2), device=dev) * 2 - 1
        for _ in range(30):
            venv.step(act)
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            venv.step(act)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / n * 1e6
        print("| %d x %d | %s | %.1f | %.3g |" % (N, B, mode, us, B / us * 1e6), flush=True)
        del env, venv
        torch.cuda.empty_cache()
