"""Step rate of a user scenario three ways (tests/plugins/ring_patrol*_env.py, 5 agents):
  per-agent file     the reference-style plugin through the callback adapter (physics on the GPU, callbacks on the host)
  tensor contract    the same scenario on device tensors (formation_gym/tensor_scenario.py), single steps and K-step calls
Usage: python3 profiles/r05_tensor_scenario.py > profiles/r05_tensor_scenario.md"""
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-formation_amd"))
import formation_gym  # noqa: E402
from formation_gym.vec_env import FormationVecEnv  # noqa: E402

PLUG = os.path.join(ROOT, "tests", "plugins")
N = 5


def make(name, B):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return formation_gym.make_env(os.path.join(PLUG, name), False, N, num_envs=B, device="cuda:0", episode_length=25)


def rate(fn, steps_per_call, min_s=1.0):
    fn(); torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < min_s:
        fn(); n += 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * steps_per_call)


print("| scenario path | envs | reset | us / step | env-steps / s |")
print("|---|---|---|---|---|")
for B in (64, 1024):
    v = FormationVecEnv(make("ring_patrol_env.py", B), reset_mode="host")
    v.reset()
    act = torch.rand((B, N, 2), device="cuda") * 2 - 1
    s = rate(lambda: v.step(act), 1, 2.0)
    print("| per-agent file (callback adapter) | %d | host | %.0f | %.3g |" % (B, s * 1e6, B / s))
for B in (1024, 16384, 262144):
    for exact in (True, False):
        e = make("ring_patrol_tensor_env.py", B)
        e.scenario.exact_reset = exact
        if exact and B > 16384:
            continue
        v = FormationVecEnv(e, reset_mode="device")
        v.reset()
        act = torch.rand((B, N, 2), device="cuda") * 2 - 1
        s = rate(lambda: v.step(act), 1)
        print("| tensor contract, single steps | %d | %s | %.0f | %.3g |" % (B, "host streams" if exact else "device generator", s * 1e6, B / s))
        if not exact:
            acts = torch.rand((25, B, N, 2), device="cuda") * 2 - 1
            s = rate(lambda: v.rollout(acts), 25)
            print("| tensor contract, 25 steps per call | %d | device generator | %.0f | %.3g |" % (B, s * 1e6, B / s))
            loop = v.capture(lambda obs: torch.tanh(4.0 * obs[..., 2:4]), 25)
            s = rate(loop.replay, 25)
            print("| tensor contract, policy + 25 steps as one hipGraph | %d | device generator | %.0f | %.3g |" % (B, s * 1e6, B / s))
