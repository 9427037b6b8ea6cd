"""Audit of the dispatch rules that depend on FgParams.obs_placed: every shape on its PLACED buffer and on an ORDINARY allocation, each
with the hint as the library sets it and with the hint inverted.  A rule is wrong where the inverted hint is faster.
   python3 profiles/r05_hint_audit.py > profiles/r05_hint_audit.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch, formation_gym
from formation_gym import placement
dev = "cuda:0"
real_is_placed = placement.is_placed
def rate(fn, byts):
    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end:
        fn(); torch.cuda.synchronize()
    reps = max(3, int(3e3 / max(1.0, byts / 6e6)))
    blocks = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        blocks.append(e0.elapsed_time(e1) / reps)
    return sorted(blocks)[2]
print("# us/step: [placed buffer: hint 1 (as shipped) | hint 0]  [ordinary buffer: hint 0 (as shipped) | hint 1]")
CLOSED = "--closed" in sys.argv           # the closed loop (env.rollout_policy) instead of the open loop
PER = {27: 3, 25: 5, 32: 2, 9: 3, 16: 4, 8: 2, 64: 4, 81: 3}
SHAPES = ((27, 4096, 20), (27, 2560, 32), (27, 8192, 10), (27, 16384, 5), (25, 4096, 20), (32, 4096, 20), (64, 2048, 20), (81, 2048, 20),
          (125, 1024, 16), (243, 2048, 4), (16, 8192, 24), (9, 4096, 128), (8, 4096, 160))
if CLOSED:
    SHAPES = ((27, 4096, 20), (27, 8192, 10), (27, 2560, 32), (25, 4096, 20), (25, 8192, 10), (32, 4096, 20), (32, 8192, 10), (16, 8192, 24), (64, 2048, 20))
for N, B, K in SHAPES:
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
    env.scenario.reset_device(env.world, rng_offset=1)
    env.auto_reset = True
    acts = (torch.rand((K, B, N, 2), device=dev) * 2 - 1).contiguous()
    real = ((24 * N * N + 17 * N) + (40 * N + 16) / K) * B
    row = []
    for label, cand in (("placed", 8), ("ordinary", 1)):
        placement.is_placed = real_is_placed
        out = env.alloc_rollout_buffers(K, candidates=cand, policy=CLOSED)
        for forced in ((1, 0) if label == "placed" else (0, 1)):
            placement.is_placed = (lambda address, v=forced: bool(v))
            env._roll_launchers.clear()
            fn = (lambda: env.rollout_policy(K, PER[N], out=out)) if CLOSED else (lambda: env.rollout(acts, out=out))
            row.append("%.2f" % (rate(fn, real) / K * 1e3))
        del out
    placement.is_placed = real_is_placed
    print("%d x %d x %d: placed [%s | %s]  ordinary [%s | %s]" % ((N, B, K) + tuple(row)), flush=True)
    env.close(); del env, acts
    torch.cuda.empty_cache()
