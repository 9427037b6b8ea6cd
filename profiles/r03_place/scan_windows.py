#!/usr/bin/env python3
"""[study script: written against the study-time arena API (fg_arena_create with an initial mapping, fg_arena_view,
fg_arena_keep; library of commits 77e8adc ... f312e76) - the shipped API maps one candidate at a time, include/formation_hip.h]
Is the rate a property of the physical REGION?  One big allocation; the rollout launch timed on windows of it at a
stride of a quarter window.   python profiles/r03_place_scan2.py N B K total_GB"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402

N, B, K, GB = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
dev = "cuda:0"
env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
env.scenario.reset_device(env.world, rng_offset=3)
env.auto_reset = True
acts = torch.zeros((K, B, N, 2), device=dev)
small = dict(reward=torch.empty((K, B, N), device=dev), indiv=torch.empty((K, B, N), device=dev),
             done=torch.zeros((K, B, N), dtype=torch.uint8, device=dev))
nfl = K * B * N * 6 * N
bytes_launch = (24 * N * N + 53 * N + 16) * B * K


def time_buffer(obs, reps=4):
    out = dict(small, obs=obs)
    env.rollout(acts, out=out)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for r in range(reps):
        env.rollout(acts, out=out)
        e[r + 1].record()
    torch.cuda.synchronize()
    env._roll_launchers.clear()
    ms = sorted(e[r].elapsed_time(e[r + 1]) for r in range(reps))
    return ms[len(ms) // 2]


total = int(GB * 1e9 / 4)
big = torch.empty(total, dtype=torch.float32, device=dev)
print("N=%d B=%d K=%d window %.2f GB inside ONE allocation of %.0f GB at %#x" % (N, B, K, nfl * 4 / 1e9, GB, big.data_ptr()))
stride = (nfl // 4) & ~3
off = 0
while off + nfl <= total:
    ms = time_buffer(big[off:off + nfl].view(K, B, N, 6 * N))
    print("  window at %7.2f GB  %.4f ms  %.0f GB/s %s" % (off * 4 / 1e9, ms, bytes_launch / (ms * 1e-3) / 1e9, "FAST" if bytes_launch / (ms * 1e-3) / 1e9 > 5700 else ""))
    off += stride
