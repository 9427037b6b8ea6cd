#!/usr/bin/env python3
"""FormationVecEnv.step (the SubprocVecEnv / DummyVecEnv replacement, train/maddpg-v2/utils/env_wrappers.py) in its reset
modes, and the captured step loop (FormationVecEnv.capture) with the built-in controller in the loop: us per vec-env step,
episodes of 100 steps whose phases are spread over the batch (about 1 % of the envs restart in EVERY step)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402
from formation_gym.vec_env import FormationVecEnv         # noqa: E402

dev = "cuda:0"
SKIP_HOST = os.environ.get("FG_SKIP_HOST", "1") == "1"
print("| shape | reset mode | us per vec-env step | env-steps/s | vs 'device' |")
print("|---|---|---|---|---|")
for N, B in ((9, 4096), (27, 4096), (27, 256), (81, 2048)):
    base = None
    for mode in ("device", "device_mt") + (() if SKIP_HOST else ("host",)):
        env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
        env.seed(1)
        venv = FormationVecEnv(env, reset_mode=mode)
        venv.reset()
        env.world.step_count.copy_((torch.arange(B, device=dev) % 100).int())     # episodes end at different steps
        if mode == "device_mt":
            venv.ts[:] = env.world.step_count.cpu().numpy()
        act = torch.rand((B, N, 2), device=dev) * 2 - 1
        for _ in range(30):
            venv.step(act)
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            venv.step(act)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / n * 1e6
        base = base or us
        print("| %d x %d | %s | %.1f | %.3g | %.2f x |" % (N, B, mode, us, B / us * 1e6, us / base), flush=True)
        del env, venv
        torch.cuda.empty_cache()

print()
print("| shape | policy in the loop | launch by launch, us/step | captured loop (20 steps per replay), us/step | speed-up |")
print("|---|---|---|---|---|")
gen = torch.Generator(device=dev); gen.manual_seed(0)
for N, B in ((27, 4096), (27, 256), (9, 4096), (9, 256), (81, 2048)):
    W = (torch.rand((6 * N, 2), generator=gen, device=dev) - 0.5) * 0.1
    for name, fn in (("get_action_BFS(ezpolicy) (fg_policy_bfs)", lambda o, out=None: formation_gym.get_action_BFS(formation_gym.ezpolicy, o, 3, out=out)),
                     ("tanh(obs @ W) (torch)", lambda o: torch.tanh(o @ W))):
        res = []
        for captured in (False, True):
            env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
            env.seed(1)
            venv = FormationVecEnv(env, reset_mode="device")
            venv.reset()
            env.world.step_count.copy_((torch.arange(B, device=dev) % 100).int())
            T = 20
            if captured:
                loop = venv.capture(fn, T)
                for _ in range(3):
                    loop.replay()
                torch.cuda.synchronize()
                reps = 15
                t0 = time.perf_counter()
                for _ in range(reps):
                    loop.replay()
                torch.cuda.synchronize()
                res.append((time.perf_counter() - t0) / (reps * T) * 1e6)
            else:
                obs = env._out["obs"]
                for _ in range(30):
                    obs = venv.step(fn(obs).contiguous())[0]
                torch.cuda.synchronize()
                n = 300
                t0 = time.perf_counter()
                for _ in range(n):
                    obs = venv.step(fn(obs).contiguous())[0]
                torch.cuda.synchronize()
                res.append((time.perf_counter() - t0) / n * 1e6)
            del env, venv
            torch.cuda.empty_cache()
        print("| %d x %d | %s | %.1f | %.1f | %.2f x |" % (N, B, name, res[0], res[1], res[0] / res[1]), flush=True)
