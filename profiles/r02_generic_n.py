#!/usr/bin/env python3
"""Single-step launches (fg_step_hd) over agent counts OUTSIDE the specialised set {3, 9, 27, 81, 243}: the run-time-N
instantiations of fg::step_kernel (flat observation writer).  HIP events around 100 queued launches, batch sized for
~300 MB of observation per step so the launch is store-bound, not launch-bound.  The K-step call (fg_rollout_hd) at
these agent counts runs the same kernel with its K-loop (one launch, no producer / writer pipeline), so this is about its rate per step as well."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402
from formation_gym import _native                         # noqa: E402

if os.environ.get("FG_EXPERIMENT_LIB"):                   # A/B runs of experiment builds (this script only)
    _native.LIB_PATH = os.path.abspath(os.environ["FG_EXPERIMENT_LIB"])

dev = "cuda:0"


def timed(fn, reps=100, warm=10):
    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end:
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


counts = sys.argv[1:] or [3, 4, 5, 8, 9, 12, 16, 20, 27, 32, 40, 50, 64, 65, 81, 100, 128, 200, 243, 256, 400, 512, 1024]
print("| agents | envs | MB per step | us per launch | algorithmic GB/s | % of 8 TB/s |")
print("|---|---|---|---|---|---|")
for item in counts:                                        # "N" (batch sized for ~300 MB per step) or "N:B"
    N = int(str(item).split(":")[0])
    per_env = 24 * N * N + 53 * N + 16
    B = max(64, min(1 << 20, int(300e6 // per_env)))
    B -= B % 64
    if ":" in str(item):
        B = int(item.split(":")[1])
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
    env.scenario.reset_device(env.world, rng_offset=1)
    env.auto_reset = True
    act = torch.rand((B, N, 2), device=dev) * 2 - 1
    us = timed(lambda: env.step(act))
    gbs = per_env * B / us / 1e3
    print("| %d | %d | %.0f | %.2f | %.0f | %.1f |" % (N, B, per_env * B / 1e6, us, gbs, gbs / 80), flush=True)
    del env, act
    torch.cuda.empty_cache()
