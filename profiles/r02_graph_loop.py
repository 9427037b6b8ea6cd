#!/usr/bin/env python3
"""A launch-bound caller loop - a small device-side policy (one matmul + tanh, torch) and env.step, T = 32 steps -
launch by launch against the same loop captured once in a hipGraph and replayed (torch.cuda.CUDAGraph).  The library
is capturable as is: it only enqueues kernels on the caller's stream.  us per env step, HIP events, median of 20."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402

dev, T = "cuda:0", 32


def med_us(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / T)
    return sorted(ts)[len(ts) // 2]


print("| shape | policy | launch by launch, us/step | hipGraph replay, us/step |")
print("|---|---|---|---|")
for N, B in ((3, 256), (9, 256), (9, 4096), (27, 256), (27, 1024), (27, 4096), (81, 256)):
    for policy in ("linear map (torch matmul + tanh)", "built-in controller (fg_policy_bfs)"):
        env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
        env.seed(1); env.reset()
        W = (torch.rand((6 * N, 2), device=dev) - 0.5) * 0.2

        def loop():
            obs = env._out["obs"]
            for _ in range(T):
                act = torch.tanh(obs @ W) if policy[0] == "l" else formation_gym.get_action_BFS(formation_gym.ezpolicy, obs, 3)
                obs = env.step(act)[0]

        eager = med_us(loop)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            loop()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            loop()
        graph = med_us(g.replay)
        print("| %d x %d | %s | %.1f | %.1f |" % (N, B, policy, eager, graph), flush=True)
        del env, g
