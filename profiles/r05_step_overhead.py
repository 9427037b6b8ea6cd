#!/usr/bin/env python3
"""Host cost of a single-step launch loop: us per env.step at 9 / 27 / 81 agents x 4096 (2048) envs through (a) the bound launcher
bench.py times (`Scenario.bind_step`: since round 5 a library-side plan, one two-argument call per step), (b) the public
`env.step(act)` with a batched action tensor, (c) the same loop captured as a hipGraph of 20 steps and replayed
(`FormationVecEnv.capture`-style, here with pre-staged actions).   python3 profiles/r05_step_overhead.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402
from formation_gym import _native                         # noqa: E402

if os.environ.get("FG_EXPERIMENT_LIB"):                   # A/B runs of experiment builds
    _native.LIB_PATH = os.path.abspath(os.environ["FG_EXPERIMENT_LIB"])
SHAPES = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(9, 4096), (27, 4096), (81, 2048)]

dev = "cuda:0"
print("| agents x envs | bound launcher us/step | env.step(act) us/step | hipGraph of 20 steps us/step |")
print("|---|---|---|---|")
for N, B in SHAPES:
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
    env.seed(1); env.reset(); env.auto_reset = True
    P = 20
    acts = (torch.rand((P, B, N, 2), device=dev) * 2 - 1).contiguous()
    launchers = [env.scenario.bind_step(env.world, acts[i], env._out, auto_reset=True) for i in range(P)]

    def rate(fn, n):
        t_end = time.perf_counter() + 0.2
        while time.perf_counter() < t_end:
            fn(); torch.cuda.synchronize()
        res = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            res.append(max(e0.elapsed_time(e1) * 1e3, (time.perf_counter() - t0) * 1e6) / (20 * n))
        return sorted(res)[2]

    t_bound = rate(lambda: [launchers[t](t + 1) for t in range(P)], P)
    t_step = rate(lambda: [env.step(acts[t]) for t in range(P)], P)
    env.use_device_rng_counter(True)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for t in range(P):
            env.step(acts[t])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for t in range(P):
                env.step(acts[t])
    t_graph = rate(lambda: g.replay(), P)
    print("| %d x %d | %.2f | %.2f | %.2f |" % (N, B, t_bound, t_step, t_graph), flush=True)
    del g, env, launchers
    torch.cuda.empty_cache()
