#!/usr/bin/env python3
"""The landmark scenarios at the reference's own shapes, 65536 envs, K-step rollout launches into a buffer beyond the
Infinity Cache (> 1 GB, so the observation stream goes to HBM): the one-env-per-lane kernels (fg_scn_lane_kernel.hpp)
against the run-time-count kernel (FgScenario.variant = 1), interleaved rounds.
   python3 profiles/r04_scenario_rollout.py [B]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402
from formation_gym import _native                         # noqa: E402

if os.environ.get("FG_EXPERIMENT_LIB"):                   # A/B runs of experiment builds
    _native.LIB_PATH = os.path.abspath(os.environ["FG_EXPERIMENT_LIB"])

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
print("# Landmark scenarios, %d envs: K-step rollout launches, observation buffer > 1 GB, device auto-reset on (one MI355X)\n" % B)
print("| scenario | agents | K | obs buffer GB | run-time-count kernel us/step | obs TB/s | lane kernel us/step | obs TB/s | of 8 TB/s | speed-up | all bytes TB/s | lane kernel without the observation stream us/step | buffer |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for scenario, N in (("basic_formation_env", 3), ("formation_hd_partial_env", 5), ("formation_hd_partial_range_env", 4),
                    ("formation_hd_obs_env", 4)):
    env = formation_gym.make_env(scenario, False, N, num_envs=B, device=dev)
    env.seed(1)
    env.scenario.reset_device(env.world, rng_offset=999)
    env.auto_reset = True
    D = env._out["obs"].shape[-1]
    K = 20
    while K * B * N * D * 4 < 1.1e9:
        K += 20
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    acts = (torch.rand((K, B, N, 2), generator=gen, device=dev) * 2 - 1).contiguous()
    f = dict(dtype=torch.float32, device=dev)
    if os.environ.get("FG_PLAIN_BUFFERS"):
        out = dict(obs=torch.empty((K, B, N, D), **f), reward=torch.empty((K, B, N), **f), indiv=torch.empty((K, B, N), **f),
                   done=torch.zeros((K, B, N), dtype=torch.uint8, device=dev))
    else:
        out = env.alloc_rollout_buffers(K)                 # the observation buffer placed (what env.rollout(acts) does by default)
    out1 = dict(out, obs=torch.empty((1, B, N, D), **f))   # obs_every = K: one observation per launch - the compute without the stream

    def run(variant, reps, o=None, every=1):
        o = out if o is None else o
        env.scenario.kernel_variant = variant
        env._roll_launchers.clear()
        env.rollout(acts, out=o, obs_every=every)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            env.rollout(acts, out=o, obs_every=every)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps / K * 1e3

    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        run(0, 2)
    res = {0: [], 1: []}
    for _ in range(5):
        for variant in (1, 0):
            res[variant].append(run(variant, 5))
    us = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    byts = B * N * D * 4
    allb = byts + B * N * (8 + 4 + 4 + 1)                  # + actions read, reward, individual reward, done written
    noobs = sorted(run(0, 5, out1, K) for _ in range(3))[1]
    print("| %s | %d | %d | %.2f | %.2f | %.2f | %.2f | %.2f | %.3f | %.2f x | %.2f | %.2f | %s |" % (
        scenario, N, K, K * byts / 1e9, us[1], byts / us[1] / 1e6, us[0], byts / us[0] / 1e6, byts / us[0] / 1e6 / 8.0, us[1] / us[0],
        allb / us[0] / 1e6, noobs, (env.placement or {}).get("kept")), flush=True)
    assert torch.isfinite(env.world.pos_x).all()
    del env, out, acts
    torch.cuda.empty_cache()
