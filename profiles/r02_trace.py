#!/usr/bin/env python3
"""HISTORICAL (round 2): the FG_TRACE stamps this script read were removed from the kernels in round 4 (the result is in
profiles/r02_step/timeline_trace.txt and DESIGN notes); kept for the record, it no longer runs against the current library.
In-kernel timeline of a single-step launch (diagnostic build: FG_EXTRA_FLAGS=-DFG_TRACE bash csrc/build.sh).
Every workgroup of fg::step_kernel stamps the 100 MHz realtime counter at: 0 entry, 1 state loaded, 2 physics done,
3 reward done / observation stream begins, 4 observation stores issued, 5 all stores acknowledged.
    python profiles/r02_trace.py N B        -> percentiles of each stamp relative to the earliest entry, in us"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402

N, B = int(sys.argv[1]), int(sys.argv[2])
trace = torch.zeros((1 << 16, 8), dtype=torch.int64, device="cuda")
os.environ["FG_TRACE_PTR"] = hex(trace.data_ptr())
import formation_gym                                      # noqa: E402
from formation_gym import _native                         # noqa: E402

env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
env.scenario.reset_device(env.world, rng_offset=3)
env.auto_reset = True
act = torch.rand((B, N, 2), device="cuda") * 2 - 1
cfg = _native.kernel_config(N)
wgs = -(-B // cfg["envs_per_wg"])
for it in range(6):
    env.scenario.step_batch(env.world, act, env._out, auto_reset=True, rng_offset=it)   # always launch_step (plain kernel)
    torch.cuda.synchronize()
t = trace[:wgs, :6].cpu().numpy().astype("float64")
t0 = t[:, 0].min()
us = (t - t0) / 100.0
names = ["entry", "state loaded", "physics done", "reward done", "obs stores issued", "stores acknowledged"]
print("N=%d B=%d  workgroups=%d  (T=%d E=%d)   stamps in us after the first workgroup's entry" % (N, B, wgs, cfg["threads"], cfg["envs_per_wg"]))
print("%-22s %8s %8s %8s %8s %8s" % ("stamp", "min", "p10", "median", "p90", "max"))
import numpy as np                                        # noqa: E402
for k, nme in enumerate(names):
    c = us[:, k]
    print("%-22s %8.2f %8.2f %8.2f %8.2f %8.2f" % (nme, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
d = us[:, 5] - us[:, 3]
print("store phase per workgroup (reward done -> acknowledged): median %.2f us, p90 %.2f us" % (np.median(d), np.percentile(d, 90)))
print("bytes per workgroup %.0f -> per-workgroup rate %.2f GB/s, x %d resident = %.0f GB/s" % (
    cfg["envs_per_wg"] * 24.0 * N * N, cfg["envs_per_wg"] * 24.0 * N * N / (np.median(d) * 1e-6) / 1e9, wgs,
    wgs * cfg["envs_per_wg"] * 24.0 * N * N / (np.median(d) * 1e-6) / 1e9))
