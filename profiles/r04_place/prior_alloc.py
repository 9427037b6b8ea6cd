#!/usr/bin/env python3
"""Does the probe's gain survive what a process has allocated BEFORE it places its buffer?  One fresh process per line:
   python3 profiles/r04_place/prior_alloc.py PRE_GB ARENA_GB [N B K]
PRE_GB of ordinary torch allocations (1 GiB pieces, kept alive) are made first - a trainer's networks and replay buffer -
then the env places its rollout buffer from an arena capped at ARENA_GB (0 = the default rule) and the launch is timed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402

PRE, ARENA = float(sys.argv[1]), float(sys.argv[2])
N, B, K = (int(x) for x in sys.argv[3:6]) if len(sys.argv) > 5 else (27, 4096, 20)
dev = torch.device("cuda:0")
env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
env.scenario.reset_device(env.world, rng_offset=1)
env.auto_reset = True
acts = (torch.rand((K, B, N, 2), device=dev) * 2 - 1).contiguous()
pre = [torch.empty(1 << 28, dtype=torch.float32, device=dev) for _ in range(int(PRE))]
if PRE % 1:
    pre.append(torch.empty(int((PRE % 1) * (1 << 28)), dtype=torch.float32, device=dev))
out = env.alloc_rollout_buffers(K, max_arena_bytes=int(ARENA * (1 << 30)) if ARENA else None)


def rate(o):
    for _ in range(10):
        env.rollout(acts, out=o)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    ev[0].record()
    for r in range(40):
        env.rollout(acts, out=o)
        ev[r + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[r].elapsed_time(ev[r + 1]) for r in range(40))
    return ts[len(ts) // 2] / K * 1e3


t_placed = rate(out)
f = dict(dtype=torch.float32, device=dev)
plain = dict(out, obs=torch.empty((K, B, N, 6 * N), **f))
t_plain = rate(plain)
r = env.placement
print("prior %5.1f GB | arena %5.1f GB kept %-26s probe %.2f s spread %s as created %.4f | placed %.2f us/step  ordinary (allocated after) %.2f" % (
    PRE, r.get("arena_GB", 0), r.get("kept"), r.get("probe_seconds", 0), r.get("spread_ms_min_median_max"), r.get("as_created_ms", 0),
    t_placed, t_plain), flush=True)
