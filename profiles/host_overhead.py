#!/usr/bin/env python3
"""Per-call cost of the Python API on top of the kernel: env.step(tensor) vs the pre-bound
launcher vs vec-env step, 27 x 4096 (kernel alone: ~16.8 us)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gym-formation_amd"))
import formation_gym  # noqa: E402
from formation_gym.vec_env import FormationVecEnv  # noqa: E402


def wall(fn, n):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    for N, B in [(27, 4096), (9, 4096)]:
        env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
        env.seed(1); env.reset()
        act = torch.rand((B, N, 2), device="cuda") * 2 - 1
        launch = env.scenario.bind_step(env.world, act, env._out, auto_reset=False)
        t_bound = wall(lambda: launch(0), 2000)
        env.world_length = 10 ** 9
        t_env = wall(lambda: env.step(act), 2000)
        venv = FormationVecEnv(env)
        venv.reset()
        t_vec = wall(lambda: venv.step(act), 2000)
        print("N=%d B=%d us per call: bound launcher %.1f | env.step %.1f | vec_env.step %.1f" % (N, B, t_bound, t_env, t_vec))


if __name__ == "__main__":
    main()
