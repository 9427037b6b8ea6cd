#!/usr/bin/env python3
"""The landmark scenarios: us per env step launched step by step (bound launchers) and as K-step rollouts
(fg_rollout_scenario, state on chip), device auto-reset on.   python3 profiles/r03_scenario_rollout.py"""
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
K = 20
print("# Landmark scenarios: one launch per step vs %d-step rollout launches (device auto-reset on, one MI355X)\n" % K)
print("| scenario | agents x envs | obs MB / step | step launches, us/step | rollout, us/step | speed-up | rollout obs GB/s |")
print("|---|---|---|---|---|---|---|")
for scenario, N, B in (("basic_formation_env", 3, 4096), ("basic_formation_env", 3, 65536),
                       ("formation_hd_partial_env", 5, 4096), ("formation_hd_partial_env", 5, 65536),
                       ("formation_hd_partial_range_env", 4, 65536), ("formation_hd_obs_env", 4, 4096),
                       ("formation_hd_obs_env", 4, 65536), ("formation_hd_obs_env", 16, 65536)):
    env = formation_gym.make_env(scenario, False, N, num_envs=B, device=dev)
    env.seed(1)
    env.reset()
    env.auto_reset = True
    D = env._out["obs"].shape[-1]
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    acts = (torch.rand((K, B, N, 2), generator=gen, device=dev) * 2 - 1).contiguous()
    f = dict(dtype=torch.float32, device=dev)
    out = dict(obs=torch.empty((K, B, N, D), **f), reward=torch.empty((K, B, N), **f), indiv=torch.empty((K, B, N), **f),
               done=torch.zeros((K, B, N), dtype=torch.uint8, device=dev))

    def steps():
        for k in range(K):
            env.step(acts[k])

    def roll():
        env.rollout(acts, out=out)

    res = []
    for fn in (steps, roll):
        t_end = time.perf_counter() + 0.2
        while time.perf_counter() < t_end:
            fn(); torch.cuda.synchronize()
        reps = 20
        blocks = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            blocks.append(e0.elapsed_time(e1) / reps / K * 1e3)
        blocks.sort()
        res.append(blocks[len(blocks) // 2])
    mb = B * N * D * 4 / 1e6
    print("| %s | %d x %d | %.1f | %.2f | %.2f | %.1f x | %.0f |" % (scenario, N, B, mb, res[0], res[1], res[0] / res[1], mb / res[1] * 1e3), flush=True)
    del env, out, acts
    torch.cuda.empty_cache()
