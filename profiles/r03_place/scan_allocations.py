#!/usr/bin/env python3
"""[study script: written against the study-time arena API (fg_arena_create with an initial mapping, fg_arena_view,
fg_arena_keep; library of commits 77e8adc ... f312e76) - the shipped API maps one candidate at a time, include/formation_hip.h]
Which allocations are fast?  M candidates of the rollout observation buffer held at once, each timed with the same
launch; then all freed and M fresh ones timed again.  Prints address, time and GB/s per candidate.
   python profiles/r03_place_scan.py N B K M"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402

N, B, K, M = (int(x) for x in sys.argv[1:5])
dev = "cuda:0"
env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
env.scenario.reset_device(env.world, rng_offset=3)
env.auto_reset = True
acts = torch.zeros((K, B, N, 2), device=dev)
small = dict(reward=torch.empty((K, B, N), device=dev), indiv=torch.empty((K, B, N), device=dev),
             done=torch.zeros((K, B, N), dtype=torch.uint8, device=dev))
nfl = K * B * N * 6 * N
bytes_launch = (24 * N * N + 53 * N + 16) * B * K


def time_buffer(obs, reps=5):
    out = dict(small, obs=obs)
    for _ in range(2):
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


print("N=%d B=%d K=%d  buffer %.1f MB, %d candidates per round" % (N, B, K, nfl * 4 / 1e6, M))
for rnd in range(3):
    held = []
    line = []
    for i in range(M):
        buf = torch.empty(nfl, dtype=torch.float32, device=dev)
        held.append(buf)
        ms = time_buffer(buf.view(K, B, N, 6 * N))
        line.append((buf.data_ptr(), ms))
    print("round %d:" % rnd)
    for p, ms in line:
        print("   %#16x  (mod 1 GiB: %4d MiB)  %.4f ms  %.0f GB/s" % (p, (p % (1 << 30)) >> 20, ms, bytes_launch / (ms * 1e-3) / 1e9))
    # re-time the first and the fastest once more while all are still held (is the rate a property of the allocation?)
    best = min(range(M), key=lambda i: line[i][1])
    print("   again: first %.4f ms, fastest (#%d) %.4f ms" % (time_buffer(held[0].view(K, B, N, 6 * N)), best, time_buffer(held[best].view(K, B, N, 6 * N))))
    del held, buf
    torch.cuda.empty_cache()
