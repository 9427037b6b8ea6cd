#!/usr/bin/env python3
"""us per step of fg_rollout_hd for N:B:K arguments, observation buffer PLACED by the probe (formation_gym/placement.py),
median of several timed blocks.  FG_EXPERIMENT_LIB selects an experiment build of the library (one process per arm; with
the probe fresh processes agree to ~1 %, profiles/r03_placement.md).  FG_AB_DIGEST=1 also prints a checksum of the
outputs of one seeded launch, so that arms can be compared for bit-identity."""
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
    N, B, K = (int(x) for x in item.split(":"))
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
    env.scenario.reset_device(env.world, rng_offset=1)
    env.auto_reset = True
    env.world.step_count.copy_((torch.arange(B, dtype=torch.int32, device=dev) * 7) % 100)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    acts = torch.rand((K, B, N, 2), device=dev, generator=gen) * 2 - 1
    out = env.alloc_rollout_buffers(K, candidates=int(os.environ.get("FG_AB_CANDIDATES", "8")),
                                    max_arena_bytes=int(float(os.environ.get("FG_AB_ARENA_GB", "192")) * (1 << 30)))
    digest = ""
    if os.environ.get("FG_AB_DIGEST"):
        env.rollout(acts, out=out)
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for k in ("obs", "reward", "indiv", "done"):
            h.update(out[k].cpu().numpy().tobytes())
        h.update(env.world.pos_x.cpu().numpy().tobytes())
        digest = "  digest " + h.hexdigest()[:16]
    t_end = time.perf_counter() + 0.25
    while time.perf_counter() < t_end:
        env.rollout(acts, out=out)
        torch.cuda.synchronize()
    reps = max(3, int(20e3 / (K * max(1.0, 24e-6 * N * N * B / 6.0))))
    blocks = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            env.rollout(acts, out=out)
        e1.record()
        torch.cuda.synchronize()
        blocks.append(e0.elapsed_time(e1) / reps / K * 1e3)
    blocks.sort()
    us = blocks[len(blocks) // 2]
    gbs = (24 * N * N + 53 * N + 16) * B / us / 1e3   # SURVEY formula (charges the state round trip per step: > 100 % possible at 3-4 agents)
    pl = env.placement or {}
    print("%d x %d, %d steps per launch: %.2f us/step (min %.2f max %.2f)  %.0f GB/s  %.1f %%  probe %s%s" % (
        N, B, K, us, blocks[0], blocks[-1], gbs, gbs / 80, [pl.get("arena_GB"), pl.get("kept"), pl.get("probe_seconds")] + pl.get("spread_ms_min_median_max", []), digest), flush=True)
    del env, out, acts
    torch.cuda.empty_cache()
