#!/usr/bin/env python3
"""The SAME rollout launch into differently composed buffers of ONE arena, for counter passes around it:
which memory-side counters separate a fast composition of the 27 x 4096 x 20 observation buffer from a slow one?

   rocprofv3 --kernel-trace --pmc <counters> -f csv json -d <dir> -- python3 profiles/r05_place_channels.py <log.json> [N B K [arena_GB]]

Every candidate is mapped alone, launched 1 + REPS times (first touch untimed) and unmapped; the log names the candidates in
launch order, so that the profiler's per-dispatch rows (kernel trace: duration, counter collection: per-instance values) can be
joined to them by position (profiles/r05_place_channels_join.py).  All candidates run the same kernel instantiation (flagged
`placed`: rollout_kernel<27,32,512,512,16,10,0,true>), whatever their composition."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402
from formation_gym import placement                       # noqa: E402

log_path = sys.argv[1]
N, B, K = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (27, 4096, 20)
arena_gb = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
REPS = int(os.environ.get("FG_PC_REPS", "4"))
SPREAD = int(os.environ.get("FG_PC_SPREAD", "10"))
dev = torch.device("cuda:0")
env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
env.scenario.reset_device(env.world, rng_offset=1)
env.auto_reset = True
acts = (torch.rand((K, B, N, 2), device=dev) * 2 - 1).contiguous()
f = dict(dtype=torch.float32, device=dev)
small = dict(reward=torch.empty((K, B, N), **f), indiv=torch.empty((K, B, N), **f), done=torch.zeros((K, B, N), dtype=torch.uint8, device=dev))
nfl = K * B * N * 6 * N
free = torch.cuda.mem_get_info(dev)[0]
total, chunk = placement.arena_geometry(nfl * 4, free, max_arena_bytes=int(arena_gb * 1e9) if arena_gb else None)
arena = placement.Arena(total, dev, chunk)
n, W = arena.chunks, -(-nfl * 4 // arena.chunk)
stream = torch.cuda.current_stream(dev)
rnd = random.Random(int(os.environ.get("FG_PC_SEED", "0")))


placement.is_placed = lambda address: True              # every candidate, the ordinary allocations too, takes the same instantiation


def time_fn(flat):
    env.rollout(acts, out=dict(small, obs=flat.view(K, B, N, 6 * N)))
    env._roll_launchers.clear()


log = {"N": N, "B": B, "K": K, "chunk_MiB": arena.chunk >> 20, "arena_chunks": n, "buffer_chunks": W, "reps": REPS, "candidates": []}


def run(label, idx):
    assert len(idx) == W and len(set(idx)) == W and max(idx) < n, (label, idx)
    addr = arena.map(idx)
    arena.kept_range = (addr, addr + W * arena.chunk)
    ms = placement._time_launch(time_fn, arena.floats(addr, nfl), stream, REPS)
    stream.synchronize()
    arena.kept_range = (0, 0)
    arena.unmap(addr)
    log["candidates"].append({"label": label, "chunks": idx, "event_ms": ms, "launches": REPS + 1})
    print("%-40s %.4f ms  %.2f us/step" % (label, ms, ms * 1e3 / K), flush=True)


run("warm-up", list(range(W)))
run("warm-up", list(range(W)))
run("neighbours (as created)", list(range(W)))
run("neighbours, middle of the arena", [n // 2 + j for j in range(W)])
run("neighbours, end of the arena", [n - W + j for j in range(W)])
for t in range(SPREAD):
    idx = sorted({min(n - 1, int((j + rnd.random()) * n / W)) for j in range(W)})
    while len(idx) < W:
        c = rnd.randrange(n)
        if c not in idx:
            idx.append(c)
    rnd.shuffle(idx)
    run("spread, shuffled #%d" % t, idx)
for R in (2, 3, 4):
    per = -(-W // R)
    if n // R >= per:
        starts = [r * (n // R) for r in range(R)]
        run("regions round-robin (%d)" % R, [starts[k % R] + k // R for k in range(W)])
        run("regions one after the other (%d)" % R, sorted(starts[k % R] + k // R for k in range(W)))
run("neighbours (as created), again", list(range(W)))
for t in range(3):                                        # ordinary allocations (held: a freed one would come straight back)
    held = log.setdefault("_held", [])
    buf = torch.empty(nfl, **f)
    held.append(buf)
    ms = placement._time_launch(time_fn, buf, stream, REPS)
    stream.synchronize()
    log["candidates"].append({"label": "ordinary allocation #%d" % t, "chunks": [], "event_ms": ms, "launches": REPS + 1})
    print("%-40s %.4f ms  %.2f us/step" % ("ordinary allocation #%d" % t, ms, ms * 1e3 / K), flush=True)
log.pop("_held")
with open(log_path, "w") as fh:
    json.dump(log, fh)
