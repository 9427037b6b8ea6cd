#!/usr/bin/env python3
"""us per single-step launch (env.step, fg_step_hd) for N:B arguments with the per-step output buffer PLACED
(env.place_step_buffers), and us per step of the closed-loop rollout (env.rollout_policy) into placed buffers for N:B:K
arguments.  FG_EXPERIMENT_LIB selects an experiment build of the library."""
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
    parts = [int(x) for x in item.split(":")]
    N, B = parts[:2]
    K = parts[2] if len(parts) > 2 else 0
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
    env.scenario.reset_device(env.world, rng_offset=1)
    env.auto_reset = True
    env.world.step_count.copy_((torch.arange(B, dtype=torch.int32, device=dev) * 7) % 100)
    if K:
        out = env.alloc_rollout_buffers(K, policy=True)
        fn = lambda: env.rollout_policy(K, 3, out=out)
        per = K
    else:
        env.place_step_buffers()
        act = torch.rand((B, N, 2), device=dev) * 2 - 1
        fn = lambda: env.step(act)
        per = 1
    t_end = time.perf_counter() + 0.25
    while time.perf_counter() < t_end:
        fn()
        torch.cuda.synchronize()
    reps = max(5, int(20e3 / (per * max(1.0, 24e-6 * N * N * B / 6.0))))
    blocks = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        blocks.append(e0.elapsed_time(e1) / reps / per * 1e3)
    blocks.sort()
    us = blocks[len(blocks) // 2]
    gbs = (24 * N * N + 53 * N + 16) * B / us / 1e3
    print("%d x %d %s: %.2f us/step (min %.2f max %.2f)  %.0f GB/s  %.1f %%  placement %s" % (
        N, B, ("closed loop, %d steps per launch" % K) if K else "single-step launches", us, blocks[0], blocks[-1], gbs, gbs / 80,
        (env.placement or {}).get("kept")), flush=True)
    del env
    torch.cuda.empty_cache()
