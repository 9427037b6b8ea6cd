#!/usr/bin/env python3
"""us per single-step launch through BOUND launchers (scenario.bind_step, what bench.py's step mode uses: no per-call
Python argument marshalling), N:B arguments, outputs digest of one seeded step.  FG_EXPERIMENT_LIB selects an experiment
build of the library."""
import hashlib
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
    N, B = (int(x) for x in item.split(":")[:2])
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
    env.scenario.reset_device(env.world, rng_offset=1)
    env.auto_reset = True
    env.world.step_count.copy_((torch.arange(B, dtype=torch.int32, device=dev) * 7) % 100)
    env.place_step_buffers()
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    P = 8
    pool = (torch.rand((P, B, N, 2), device=dev, generator=gen) * 2 - 1).contiguous()
    out = env._out
    launch = [env.scenario.bind_step(env.world, pool[i], out, auto_reset=True) for i in range(P)]
    for t in range(3):
        launch[t](t)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for k in ("obs", "reward", "indiv", "done"):
        h.update(out[k].cpu().numpy().tobytes())
    h.update(env.world.pos_x.cpu().numpy().tobytes())
    h.update(env.world.step_count.cpu().numpy().tobytes())
    t_end = time.perf_counter() + 0.25
    t = 3
    while time.perf_counter() < t_end:
        for _ in range(64):
            launch[t % P](t); t += 1
        torch.cuda.synchronize()
    reps = max(50, int(20e3 / max(1.0, 24e-6 * N * N * B / 6.0)))
    blocks = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            launch[t % P](t); t += 1
        e1.record()
        torch.cuda.synchronize()
        blocks.append(e0.elapsed_time(e1) / reps * 1e3)
    blocks.sort()
    us = blocks[len(blocks) // 2]
    gbs = (24 * N * N + 53 * N + 16) * B / us / 1e3
    print("%d x %d single-step launches: %.2f us/step (min %.2f max %.2f)  %.0f GB/s  %.1f %%  digest %s" % (
        N, B, us, blocks[0], blocks[-1], gbs, gbs / 80, h.hexdigest()[:16]), flush=True)
    del env, launch, out
    torch.cuda.empty_cache()
