#!/usr/bin/env python3
"""Which chunk selections are fast?  One arena (chunks placed in index order), the 27 x 4096 x 20 rollout buffer (11 chunks
of 128 MiB) composed by RULE instead of by chance: regular strides, residues modulo small periods, clusters, orders.
us per step of the env's own 20-step launch, median of 12 launches per selection.
   python3 profiles/r03_place/selection_rules.py [N B K]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402
from formation_gym import placement                       # noqa: E402

N, B, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (27, 4096, 20)
dev = torch.device("cuda:0")
env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
env.scenario.reset_device(env.world, rng_offset=1)
env.auto_reset = True
acts = (torch.rand((K, B, N, 2), device=dev) * 2 - 1).contiguous()
f = dict(dtype=torch.float32, device=dev)
small = dict(reward=torch.empty((K, B, N), **f), indiv=torch.empty((K, B, N), **f), done=torch.zeros((K, B, N), dtype=torch.uint8, device=dev))
nfl = K * B * N * 6 * N
free = torch.cuda.mem_get_info(dev)[0]
total, chunk = placement.arena_geometry(nfl * 4, free)
arena = placement.Arena(total, dev, chunk)
n, W = arena.chunks, -(-nfl * 4 // arena.chunk)
stream = torch.cuda.current_stream(dev)
print("arena %d chunks of %d MiB, buffer %d chunks (%d x %d x %d)" % (n, arena.chunk >> 20, W, N, B, K), flush=True)


def time_fn(flat):
    env.rollout(acts, out=dict(small, obs=flat.view(K, B, N, 6 * N)))
    env._roll_launchers.clear()


def run(label, idx, placed=True):
    assert len(idx) == W and len(set(idx)) == W and max(idx) < n, (label, idx)
    addr = arena.map(idx)
    arena.kept_range = (addr, addr + W * arena.chunk) if placed else (0, 0)
    ms = placement._time_launch(time_fn, arena.floats(addr, nfl), stream, 12)
    stream.synchronize()
    arena.kept_range = (0, 0)
    arena.unmap(addr)
    print("%-58s %.2f us/step   chunks %s" % (label, ms * 1e3 / K, idx if W <= 16 else idx[:8] + ["..."]), flush=True)
    return ms


for _ in range(20):                                           # clocks up
    run("warm-up", list(range(W)))
    break
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    a = arena.map(list(range(W))); placement._time_launch(time_fn, arena.floats(a, nfl), stream, 4); stream.synchronize(); arena.unmap(a)
print("-- neighbours, with and without the writer geometry of a placed buffer")
run("neighbours 0..W-1 (not flagged placed)", list(range(W)), placed=False)
run("neighbours 0..W-1 (flagged placed)", list(range(W)))
run("neighbours in the middle of the arena", [n // 2 + j for j in range(W)])
print("-- regular stride s, ascending order")
for s in (2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, (n - 1) // (W - 1)):
    if s * (W - 1) < n:
        run("stride %d (span %.1f GB)" % (s, s * (W - 1) * arena.chunk / 1e9), [s * j for j in range(W)])
smax = (n - 1) // (W - 1)
print("-- full span (stride %d): order of the chunks inside the buffer" % smax)
asc = [smax * j for j in range(W)]
run("ascending", asc)
run("descending", asc[::-1])
g = 7 if W == 11 else max(1, int(round(W * 0.618)))
run("golden stride order", [asc[(k * g) % W] for k in range(W)])
run("interleaved halves", [asc[j // 2 + (W + 1) // 2 * (j % 2)] for j in range(W)])
print("-- full span, residues: all chunk indices = r (mod P)")
for P in (2, 4, 8, 16, 32, 64):
    s = (smax // P) * P
    if s >= P:
        run("indices = 0 mod %d (stride %d)" % (P, s), [s * j for j in range(W)])
        run("indices = j mod %d (stride %d + 1)" % (P, s), [(s + 1) * j for j in range(W)] if (s + 1) * (W - 1) < n else [s * j + (j % P) for j in range(W)])
print("-- clusters")
h = W // 2
run("two clusters at the ends", list(range(h)) + [n - 1 - j for j in range(W - h)])
run("two clusters, interleaved order", [x for p in zip(range(h), [n - 1 - j for j in range(h)]) for x in p] + ([n - 1 - h] if W % 2 else []))
run("three clusters", [c * (n // 3) + j for j in range(-(-W // 3)) for c in range(3)][:W])
arena.close()
