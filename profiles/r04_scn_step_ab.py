#!/usr/bin/env python3
"""Single-step launches (one env.step per launch) of the landmark scenarios: the one-env-per-lane kernel against the
run-time-count kernel over batch sizes (is the lane kernel the right choice for K = 1 at small batches too?).
   python3 profiles/r04_scn_step_ab.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402

dev = "cuda:0"
print("| scenario | envs | lane kernel us/step | run-time-count kernel us/step |")
print("|---|---|---|---|")
for scenario, N in (("basic_formation_env", 3), ("formation_hd_obs_env", 4)):
    for B in (1, 256, 4096, 16384, 65536):
        env = formation_gym.make_env(scenario, False, N, num_envs=B, device=dev)
        env.seed(1); env.scenario.reset_device(env.world, rng_offset=3); env.auto_reset = True
        acts = [(torch.rand((B, N, 2), device=dev) * 2 - 1).contiguous() for _ in range(16)]
        res = {}
        for variant in (0, 1, 0, 1):
            env.scenario.kernel_variant = variant
            env._launchers.clear()
            for k in range(50):
                env.step(acts[k % 16])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(400):
                env.step(acts[k % 16])
            e1.record(); torch.cuda.synchronize()
            res.setdefault(variant, []).append(e0.elapsed_time(e1) / 400 * 1e3)
        print("| %s | %d | %.2f | %.2f |" % (scenario, B, min(res[0]), min(res[1])), flush=True)
        del env
