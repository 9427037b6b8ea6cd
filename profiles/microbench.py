#!/usr/bin/env python3
"""Stage timings on the GPU box: physics-only, observe-only, fused step, and two
library references for the achievable store rate at the same buffer size
(torch fill_ = pure store stream; torch copy_ = load+store stream)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gym-formation_amd"))
import formation_gym  # noqa: E402


def timeit(fn, n=300, warm=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n      # us


def main():
    res = []
    for N, B in [(27, 4096), (9, 4096), (81, 2048), (243, 8192)]:
        env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
        env.seed(1); env.reset()
        act = torch.rand((B, N, 2), device="cuda") * 2 - 1
        env.world.action_u.copy_(act)
        out = env._out
        n = 300 if N < 243 else 30
        t_phys = timeit(lambda: env.world.step(), n)
        t_obs = timeit(lambda: env.scenario.observe_batch(env.world, out), n)
        launch = env.scenario.bind_step(env.world, act, out, auto_reset=False)
        env.reset()
        t_step = timeit(lambda: launch(0), n)
        # everything except the observation stream: K=1 rollout with obs_every=2 never writes obs
        seq = dict(reward=out["reward"][None], indiv=out["indiv"][None], done=out["done"][None], obs=out["obs"][None])
        act1 = act[None].contiguous()
        from formation_gym import _native
        lib = _native.load()
        w, sc = env.world, env.scenario
        P = sc.params(w)
        args = (w.num_envs, N, 1, w.pos_x.data_ptr(), w.pos_y.data_ptr(), w.vel_x.data_ptr(), w.vel_y.data_ptr(),
                act1.data_ptr(), sc.ideal_shape.data_ptr(), sc.ideal_vel.data_ptr(), w.step_count.data_ptr(),
                seq["obs"].data_ptr(), seq["reward"].data_ptr(), seq["indiv"].data_ptr(), seq["done"].data_ptr(),
                2, _native.current_stream())
        t_noobs = timeit(lambda: lib.fg_rollout_hd(P, *args), n)
        pargs = (w.num_envs, N, w.pos_x.data_ptr(), w.pos_y.data_ptr(), w.vel_x.data_ptr(), w.vel_y.data_ptr(),
                 act1.data_ptr(), _native.current_stream())
        t_phys = timeit(lambda: lib.fg_physics_step(P, *pargs), n)
        empty = torch.empty(1, device="cuda")
        t_null = timeit(lambda: empty.fill_(0.0), n)
        obs = out["obs"]
        t_fill = timeit(lambda: obs.fill_(1.0), n)
        src = torch.empty_like(obs)
        t_copy = timeit(lambda: obs.copy_(src), n)
        gb = obs.numel() * 4 / 1e9
        res.append(dict(N=N, B=B, obs_MB=round(gb * 1e3, 1), physics_us=round(t_phys, 2), observe_us=round(t_obs, 2),
                        step_us=round(t_step, 2), step_without_obs_us=round(t_noobs, 2), tiny_fill_us=round(t_null, 2), fill_us=round(t_fill, 2), fill_GBps=round(gb / t_fill * 1e6, 0),
                        copy_us=round(t_copy, 2), copy_store_GBps=round(gb / t_copy * 1e6, 0)))
        print(json.dumps(res[-1]), flush=True)
        del env, src


if __name__ == "__main__":
    main()
