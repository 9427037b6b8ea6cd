#!/usr/bin/env python3
"""Per-step launches of one 4096-env batch vs two 2048-env halves driven on two HIP streams with no
cross-stream dependency (the way an actor that alternates between two env groups runs them): the
physics of one half can overlap the store drain of the other."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gym-formation_amd"))
import formation_gym  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 27
    Bt = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    steps = 2000

    def make(B, seed):
        env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
        env.seed(seed); env.reset()
        act = torch.rand((B, N, 2), device="cuda") * 2 - 1
        return env, act

    def bound(env, act):
        return env.scenario.bind_step(env.world, act, env._out, auto_reset=True)

    whole, act_w = make(Bt, 1)
    lw = bound(whole, act_w)
    for _ in range(100):
        lw(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        lw(t)
    torch.cuda.synchronize()
    t_whole = (time.perf_counter() - t0) / steps * 1e6

    res = {}
    for parts in (2, 4):
        streams = [torch.cuda.Stream() for _ in range(parts)]
        launch = []
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                env, act = make(Bt // parts, 2 + k)
                launch.append((s, bound(env, act), env, act))          # bound to stream s
        torch.cuda.synchronize()
        for _ in range(100):
            for s, l, _, _ in launch:
                l(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(steps):
            for s, l, _, _ in launch:
                l(t)
        torch.cuda.synchronize()
        res[parts] = (time.perf_counter() - t0) / steps * 1e6
    print("N=%d B=%d: one batch %.2f us/step | 2 halves on 2 streams %.2f us per step of all envs | 4 quarters on 4 streams %.2f"
          % (N, Bt, t_whole, res[2], res[4]))


if __name__ == "__main__":
    main()
