#!/usr/bin/env python3
"""Timings of the kernels beside the headline path (HIP events, 200 launches each, one process):
the BFS controller (fg_policy_bfs / fg_policy_bfs_state), closed-loop rollouts (fg_rollout_hd_policy), the landmark
scenarios (fg_step_scenario / fg_step_basic) and the device resets.  Run it bare for the table, or under
`rocprofv3 --kernel-trace --stats` for the per-kernel CSV (profiles/r02_aux_kernel_stats.csv)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402
from formation_gym.policy_bfs import bfs_actions          # noqa: E402

dev = "cuda:0"


def timed(fn, reps=200, warm=20):
    import time
    t_end = time.perf_counter() + 0.15            # bring the clocks up first (short timings are otherwise taken on the ramp)
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
    return e0.elapsed_time(e1) / reps * 1e3               # us per call


print("| what | shape | us per launch | note |")
print("|---|---|---|---|")
for N, B in ((9, 4096), (27, 4096), (81, 2048), (243, 8192)):
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
    env.scenario.reset_device(env.world, rng_offset=1)
    obs = env._out["obs"]
    env.scenario.observe_batch(env.world, {"obs": obs, "reward": env._out["reward"]})
    act = torch.empty((B, N, 2), device=dev)
    us = timed(lambda: bfs_actions(obs, 3, out=act))
    rd = B * 24 * N + B * N * 8
    print("| fg_policy_bfs (from observation row 0) | %d x %d | %.2f | reads %.1f MB + writes %.1f MB: latency-bound |" % (N, B, us, B * 24 * N / 1e6, B * N * 8 / 1e6))
    us = timed(lambda: env.scenario.policy_actions(env.world, 3, out=act))
    print("| fg_policy_bfs_state (from pos / ideal shape) | %d x %d | %.2f | |" % (N, B, us))
    env.auto_reset = True
    us_step = timed(lambda: env.step(act), reps=100)
    print("| fg_step_hd (env.step, for scale) | %d x %d | %.2f | |" % (N, B, us_step))
    K = 20 if N < 243 else 4
    f = dict(dtype=torch.float32, device=dev)
    out = dict(obs=torch.empty((K, B, N, 6 * N), **f), reward=torch.empty((K, B, N), **f), indiv=torch.empty((K, B, N), **f),
               done=torch.zeros((K, B, N), dtype=torch.uint8, device=dev), act=torch.empty((K, B, N, 2), **f))
    us = timed(lambda: env.rollout_policy(K, 3, out=out), reps=30, warm=5)
    acts = torch.rand((K, B, N, 2), device=dev) * 2 - 1
    out2 = {k: v for k, v in out.items() if k != "act"}
    us2 = timed(lambda: env.rollout(acts, out=out2), reps=30, warm=5)
    print("| fg_rollout_hd_policy (controller in the rollout kernel), K = %d | %d x %d | %.2f per step | open loop (fg_rollout_hd): %.2f per step |" % (K, N, B, us / K, us2 / K))
    del env, out, out2, obs
    torch.cuda.empty_cache()
for scn, N in (("basic_formation_env", 3), ("formation_hd_partial_env", 4), ("formation_hd_partial_range_env", 4), ("formation_hd_obs_env", 4),
               ("formation_hd_obs_env", 16)):
    for B in (4096, 65536):
        env = formation_gym.make_env(scn, False, N, num_envs=B, device=dev)
        env.reset()
        act = torch.rand((B, N, 2), device=dev) * 2 - 1
        us = timed(lambda: env.step(act))
        D = env._out["obs"].shape[-1]
        by = B * N * (D * 4 + 8 * 4 + 4 + 1)
        print("| %s env.step (fg::scn_kernel) | %d x %d, obs dim %d | %.2f | %.1f MB per step -> %.0f GB/s |" % (scn, N, B, D, us, by / 1e6, by / (us * 1e-6) / 1e9))
        del env
for N, B in ((27, 4096), (243, 8192)):
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
    us = timed(lambda: env.scenario.reset_device(env.world, rng_offset=3))
    print("| fg_reset_hd (counter RNG, all envs) | %d x %d | %.2f | |" % (N, B, us))
    env.scenario.upload_mt_streams(env.world)
    us = timed(lambda: env.scenario.reset_mt(env.world), reps=50, warm=5)
    print("| fg_reset_hd_mt (legacy MT19937 streams on device, all envs) | %d x %d | %.2f | |" % (N, B, us))
