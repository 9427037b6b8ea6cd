#!/usr/bin/env python3
"""Placement BY RULE: an arena no larger than the buffer whose chunks come in R groups held `spacer` GB apart while the
arena is created (fg_arena_create_spread), mapped round-robin over the groups.  One fresh process per configuration:
   python3 profiles/r04_place/spread_rule.py N B K R SPACER_GB [CHUNK_MiB]
prints us/step of the env's own K-step launch into (a) an ordinary torch allocation, (b) the arena's chunks in index
order (neighbours), (c) the round-robin composition - median of 12 launches each."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402
from formation_gym import placement                       # noqa: E402

N, B, K, R = (int(x) for x in sys.argv[1:5])
S = float(sys.argv[5])
CH = int(sys.argv[6]) if len(sys.argv) > 6 else 0
MODE = int(sys.argv[7]) if len(sys.argv) > 7 else 0      # spacer: 0 hipMalloc, 1 chunks created, 2 chunks created + mapped
dev = torch.device("cuda:0")
env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
env.scenario.reset_device(env.world, rng_offset=1)
env.auto_reset = True
acts = (torch.rand((K, B, N, 2), device=dev) * 2 - 1).contiguous()
f = dict(dtype=torch.float32, device=dev)
small = dict(reward=torch.empty((K, B, N), **f), indiv=torch.empty((K, B, N), **f), done=torch.zeros((K, B, N), dtype=torch.uint8, device=dev))
nfl = K * B * N * 6 * N
nbytes = nfl * 4
stream = torch.cuda.current_stream(dev)
free0 = torch.cuda.mem_get_info(dev)[0]


def time_fn(flat):
    env.rollout(acts, out=dict(small, obs=flat.view(K, B, N, 6 * N)))
    env._roll_launchers.clear()


plain = torch.empty(nfl, **f)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:                        # clocks up
    placement._time_launch(time_fn, plain, stream, 4)
us_plain = placement._time_launch(time_fn, plain, stream, 12) * 1e3 / K

chunk = (CH << 20) if CH else 0
if not chunk:
    chunk = 32 << 20
    while chunk < (1 << 30) and nbytes // chunk > 16:
        chunk <<= 1
t0 = time.perf_counter()
arena = placement.Arena(nbytes, dev, chunk, regions=R | (MODE << 24), spacer_bytes=int(S * (1 << 30)))
t_create = time.perf_counter() - t0
n, W = arena.chunks, -(-nbytes // arena.chunk)
assert n == W


def run(idx, placed):
    addr = arena.map(idx)
    arena.kept_range = (addr, addr + W * arena.chunk) if placed else (0, 0)
    ms = placement._time_launch(time_fn, arena.floats(addr, nfl), stream, 12)
    stream.synchronize()
    arena.kept_range = (0, 0)
    arena.unmap(addr)
    return ms * 1e3 / K


groups = [list(range(n * r // R, n * (r + 1) // R)) for r in range(R)]
rr = [g[k] for k in range(max(len(g) for g in groups)) for g in groups if k < len(g)]
us_seq_p = run(list(range(W)), True)
us_rr_p = run(rr, True)
us_rr_u = run(rr, False)
us_rr_p2 = run(rr, True)
free1 = torch.cuda.mem_get_info(dev)[0]
arena.close()
print("mode %d " % MODE + "N %d B %d K %d buffer %.2f GB chunk %d MiB x %d | R %d spacer asked %.0f GB held %.1f GB create %.2f s | plain alloc %.2f | "
      "index order (placed flag) %.2f | round-robin placed %.2f / %.2f  unflagged %.2f us/step | free before %.1f GB, with arena %.1f GB"
      % (N, B, K, nbytes / 1e9, arena.chunk >> 20, W, R, S, arena.spacer_held / 1e9, t_create, us_plain, us_seq_p, us_rr_p, us_rr_p2,
         us_rr_u, free0 / 1e9, free1 / 1e9), flush=True)
