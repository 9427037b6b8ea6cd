#!/usr/bin/env python3
"""K-step rollout launches at the agent counts of the hierarchies other than 3^L (per_layer 2, 4, 5, 8: N = 4, 8, 16, 25, 32,
64, 125), which round 4 gave pipelined kernels (round 3: the K-loop of the run-time-N step kernel, 57-66 % of peak): us per
step and fraction of the HBM peak (SURVEY formula bytes) for open-loop launches into the env's own placed buffers and for the
closed loop with the built-in controller in ONE launch.   python3 profiles/r05_generic_n.py [N:B:per[:K] ...]
Round 5: the wide kernels (81 / 125 / 243 agents) prefetch their actions one step ahead; shapes from the command line."""
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
K = 20
SHAPES = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [
    (4, 65536, 2), (8, 65536, 2), (16, 8192, 4), (16, 32768, 2), (25, 4096, 5), (25, 16384, 5), (32, 4096, 2),
    (64, 2048, 4), (64, 4096, 8), (81, 2048, 3), (125, 1024, 5), (125, 4096, 5)]
print("# Rollout launches (%d steps unless a row says otherwise, every observation written, device auto-reset), one MI355X\n" % K)
print("`of 8 TB/s` = the SURVEY formula (24 N^2 + 53 N + 16 bytes per env-step, which charges the state's round trip to every step: it")
print("passes 1 at 4 agents); `real` = the bytes a K-step launch moves, (24 N^2 + 17 N) per env-step + the state once.\n")
print("| agents x envs | obs MB/step | buffer | open loop us/step | of 8 TB/s | real | closed loop (per_layer) us/step | of 8 TB/s | real | single-step launches us/step |")
print("|---|---|---|---|---|---|---|---|---|---|")
for shape in SHAPES:
    N, B, per = shape[:3]
    K = shape[3] if len(shape) > 3 else 20
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
    env.scenario.reset_device(env.world, rng_offset=1)
    env.auto_reset = True
    acts = (torch.rand((K, B, N, 2), device=dev) * 2 - 1).contiguous()
    byts = (24 * N * N + 53 * N + 16) * B
    real = ((24 * N * N + 17 * N) + (40 * N + 16) / K) * B

    def rate(fn, reps=None):
        t_end = time.perf_counter() + 0.2
        while time.perf_counter() < t_end:
            fn(); torch.cuda.synchronize()
        reps = reps or max(3, int(4e3 / max(1.0, byts / 6e6)))
        blocks = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            blocks.append(e0.elapsed_time(e1) / reps)
        return sorted(blocks)[2]
    t_open = rate(lambda: env.rollout(acts)) / K * 1e3
    kept = (env.placement or {}).get("kept", "ordinary")
    env.close()
    t_closed = rate(lambda: env.rollout_policy(K, per)) / K * 1e3
    env.close()
    step = lambda: [env.step(acts[k]) for k in range(K)]
    t_step = rate(step, reps=3) / K * 1e3
    print("| %d x %d%s | %.1f | %s | %.2f | %.3f | %.3f | %.2f (%d) | %.3f | %.3f | %.2f |" % (
        N, B, "" if K == 20 else " (%d steps)" % K, byts / 1e6, kept, t_open, byts / t_open / 8e6, real / t_open / 8e6, t_closed, per, byts / t_closed / 8e6,
        real / t_closed / 8e6, t_step), flush=True)
    del env, acts
    torch.cuda.empty_cache()
