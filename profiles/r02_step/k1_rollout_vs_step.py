#!/usr/bin/env python3
"""One step per launch: fg_step_hd (step_kernel, rows writer) against fg_rollout_hd with K = 1 routed to the
producer / writer kernels (experiment build -DFG_ROLL_MIN_K=1, FG_EXPERIMENT_LIB).  us per launch, HIP events."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402
from formation_gym import _native                         # noqa: E402

if os.environ.get("FG_EXPERIMENT_LIB"):
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


print("| shape | MB per step | env.step us | env.rollout(K=1) us |")
print("|---|---|---|---|")
for item in sys.argv[1:]:
    N, B = (int(x) for x in item.split(":"))
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
    env.scenario.reset_device(env.world, rng_offset=1)
    env.auto_reset = True
    act = torch.rand((1, B, N, 2), device=dev) * 2 - 1
    f = dict(dtype=torch.float32, device=dev)
    out = dict(obs=env._out["obs"].view(1, B, N, 6 * N), reward=torch.empty((1, B, N), **f), indiv=torch.empty((1, B, N), **f),
               done=torch.zeros((1, B, N), dtype=torch.uint8, device=dev))
    a = timed(lambda: env.step(act[0]))
    b = timed(lambda: env.rollout(act, out=out))
    print("| %d x %d | %.0f | %.2f | %.2f |" % (N, B, (24 * N * N + 53 * N + 16) * B / 1e6, a, b), flush=True)
    del env, out, act
    torch.cuda.empty_cache()
