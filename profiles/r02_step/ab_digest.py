#!/usr/bin/env python3
"""Digest of the results of seeded step / rollout launches (crowded and sparse states, auto-reset on): run once per
library build (FG_EXPERIMENT_LIB) and compare the lines - equal digests = bit-identical results."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402
from formation_gym import _native                         # noqa: E402

if os.environ.get("FG_EXPERIMENT_LIB"):
    _native.LIB_PATH = os.path.abspath(os.environ["FG_EXPERIMENT_LIB"])
dev = "cuda:0"


def digest(*ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().contiguous().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


for N, B, K in ((3, 500, 6), (9, 700, 6), (27, 300, 6), (81, 40, 4), (243, 6, 3), (10, 200, 4), (50, 30, 4), (100, 12, 3), (300, 3, 2)):
    for crowd in (1.0, 0.3, 0.08):
        env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
        env.scenario.seed(3)
        env.scenario.reset_device(env.world, rng_offset=77)
        env.world.pos_x.mul_(crowd); env.world.pos_y.mul_(crowd)
        env.world.step_count.copy_((torch.arange(B, device=dev) % 9 + 92).int())
        env.auto_reset = True
        gen = torch.Generator(device=dev); gen.manual_seed(N)
        acts = torch.rand((K, B, N, 2), generator=gen, device=dev) * 2 - 1
        outs = []
        for k in range(K):
            o, r, d, i = env.step(acts[k])
            outs += [o.clone(), r.clone(), i["individual_reward"].clone()]
        o, r, d, i = env.rollout(acts)
        pos, vel = env.world.get_state()
        print("%d x %d crowd %.2f  steps %s  rollout %s  state %s  finite %s" % (
            N, B, crowd, digest(*outs), digest(o, r, i["individual_reward"]), digest(pos, vel), bool(torch.isfinite(o).all())))
