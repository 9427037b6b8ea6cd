#!/usr/bin/env python3
"""us per step of fg_rollout_hd (env.rollout into pre-allocated buffers) for N:B:K arguments; FG_EXPERIMENT_LIB selects an
experiment build of the library (A/B runs in one gpurun call)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402
from formation_gym import _native                         # noqa: E402

if os.environ.get("FG_EXPERIMENT_LIB"):
    _native.LIB_PATH = os.path.abspath(os.environ["FG_EXPERIMENT_LIB"])
dev = "cuda:0"
for item in sys.argv[1:]:
    N, B, K = (int(x) for x in item.split(":"))
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
    env.scenario.reset_device(env.world, rng_offset=1)
    env.auto_reset = True
    acts = torch.rand((K, B, N, 2), device=dev) * 2 - 1
    f = dict(dtype=torch.float32, device=dev)
    out = dict(obs=torch.empty((K, B, N, 6 * N), **f), reward=torch.empty((K, B, N), **f), indiv=torch.empty((K, B, N), **f),
               done=torch.zeros((K, B, N), dtype=torch.uint8, device=dev))
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        env.rollout(acts, out=out)
        torch.cuda.synchronize()
    reps = max(5, int(30e3 / (K * max(1.0, 24e-6 * N * N * B / 6.0))))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        env.rollout(acts, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps / K * 1e3
    gbs = (24 * N * N + 53 * N + 16) * B / us / 1e3
    print("%d x %d, %d steps per launch: %.2f us/step  %.0f GB/s  %.1f %%" % (N, B, K, us, gbs, gbs / 80), flush=True)
    del env, out, acts
    torch.cuda.empty_cache()
