#!/usr/bin/env python3
"""Launch 3 x 100 kernels in a fixed order (physics-only, step without the
observation stream, full step) so that a rocprofv3 --kernel-trace of this
script gives GPU-side durations per stage:  rocprofv3 --kernel-trace
--output-format csv -d OUT -- python3 profiles/stages.py [N B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gym-formation_amd"))
import formation_gym  # noqa: E402
from formation_gym import _native  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 27
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
REP = 100
env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
env.seed(1); env.reset()
w, sc, out = env.world, env.scenario, env._out
act1 = (torch.rand((1, B, N, 2), device="cuda") * 2 - 1).contiguous()
lib = _native.load()
P = sc.params(w)
st = _native.current_stream()
base = (w.pos_x.data_ptr(), w.pos_y.data_ptr(), w.vel_x.data_ptr(), w.vel_y.data_ptr(), act1.data_ptr())
roll = (B, N, 1) + base + (sc.ideal_shape.data_ptr(), sc.ideal_vel.data_ptr(), w.step_count.data_ptr(),
                           out["obs"].data_ptr(), out["reward"].data_ptr(), out["indiv"].data_ptr(),
                           out["done"].data_ptr())
torch.cuda.synchronize()
for _ in range(REP):
    lib.fg_physics_step(P, B, N, *base, st)
torch.cuda.synchronize()
env.reset()
for _ in range(REP):
    lib.fg_rollout_hd(P, *roll, 2, st)          # obs_every = 2 with K = 1: no observation written
torch.cuda.synchronize()
env.reset()
for _ in range(REP):
    lib.fg_rollout_hd(P, *roll, 1, st)          # full step
torch.cuda.synchronize()
