#!/usr/bin/env python3
"""Per-call wall time of the reference-style API (num_envs = 1, lists of per-agent NumPy arrays in and out):
what a drop-in user of `formation_gym.make_env(...).step(act_n)` sees, host <-> device copies included."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gym-formation_amd"))
import formation_gym  # noqa: E402


def main():
    for scenario, N in [("basic_formation_env", 3), ("formation_hd_env", 3), ("formation_hd_env", 9), ("formation_hd_env", 27),
                        ("formation_hd_env", 81)]:
        env = formation_gym.make_env(scenario, False, N)
        env.seed(1); env.reset()
        rs = np.random.RandomState(0)
        n, t_reset = 300, 0.0
        for _ in range(30):
            env.step([rs.uniform(-1, 1, 2) for _ in range(N)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(n):
            obs_n, rew_n, done_n, info_n = env.step([rs.uniform(-1, 1, 2) for _ in range(N)])
            if all(done_n):
                t1 = time.perf_counter(); env.reset(); t_reset += time.perf_counter() - t1
        dt = (time.perf_counter() - t0 - t_reset) / n
        print("%-22s N=%-3d reference-style env.step: %.1f us per call (%.0f env-steps/s)" % (scenario, N, dt * 1e6, 1 / dt))


if __name__ == "__main__":
    main()
