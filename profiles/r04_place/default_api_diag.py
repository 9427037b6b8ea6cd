#!/usr/bin/env python3
"""env.rollout(acts) with no buffers (placed on first use) against the same launch into an ordinary tensor, one process,
alternating arms.   python3 profiles/r04_place/default_api_diag.py [N B K]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402

N, B, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (27, 4096, 20)
dev = torch.device("cuda:0")
envs = []
for _ in range(2):
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
    env.scenario.reset_device(env.world, rng_offset=1)
    env.auto_reset = True
    envs.append(env)
a, b = envs
acts = (torch.rand((K, B, N, 2), device=dev) * 2 - 1).contiguous()
f = dict(dtype=torch.float32, device=dev)
plain = dict(obs=torch.empty((K, B, N, 6 * N), **f), reward=torch.empty((K, B, N), **f), indiv=torch.empty((K, B, N), **f),
             done=torch.zeros((K, B, N), dtype=torch.uint8, device=dev))


def rate(env, out):
    for _ in range(10):
        env.rollout(acts, out=out)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    ev[0].record()
    for r in range(40):
        env.rollout(acts, out=out)
        ev[r + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[r].elapsed_time(ev[r + 1]) for r in range(40))
    return ts[len(ts) // 2] / K * 1e3


b.rollout(acts)
print("probe:", b.placement)
for rnd in range(4):
    print("round %d: ordinary %.2f us/step   default API (placed) %.2f us/step" % (rnd, rate(a, plain), rate(b, None)), flush=True)
