#!/usr/bin/env python3
"""Rollout driver: the counterpart of the reference's `test.py` (argparse ->
make_env -> step loop with a random or the built-in hierarchical policy),
batched over `--num-envs` environments on one MI355X and without the pyglet
window.

    python gym-formation_amd/demo.py -s formation_hd_env -n 3 --num-layer 3 --num-envs 4096 --steps 300
    python gym-formation_amd/demo.py -s formation_hd_env -n 3 --num-layer 2 -r
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

import formation_gym  # noqa: E402
from formation_gym.vec_env import FormationVecEnv  # noqa: E402

if __name__ == '__main__':
    parser = argparse.ArgumentParser(description=None)
    parser.add_argument('-s', '--scenario', default='formation_hd_env', help='scenario name or path of a scenario script')
    parser.add_argument('-n', '--num-agents', type=int, default=3, help='agents per layer')
    parser.add_argument('-r', '--random', action='store_true', help='random policy instead of the BFS demo policy')
    parser.add_argument('--num-layer', type=int, default=1, help='hierarchy depth: total agents = n ** layers')
    parser.add_argument('--num-envs', type=int, default=1024)
    parser.add_argument('--steps', type=int, default=300)
    parser.add_argument('--seed', type=int, default=1)
    args = parser.parse_args()

    total = args.num_agents ** args.num_layer
    env = formation_gym.make_env(args.scenario, benchmark=False, num_agents=total,
                                 num_envs=args.num_envs, device='cuda:0')
    env.seed(args.seed)
    device_reset = args.scenario == 'formation_hd_env'
    venv = FormationVecEnv(env, reset_mode='device' if device_reset else 'host')
    obs = venv.reset()
    gen = torch.Generator(device='cuda'); gen.manual_seed(args.seed)
    ret = torch.zeros(args.num_envs, device='cuda')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.steps):
        if args.random or args.scenario != 'formation_hd_env':
            act = torch.rand((args.num_envs, total, 2), generator=gen, device='cuda') * 2 - 1
        else:
            act = formation_gym.get_action_BFS(formation_gym.ezpolicy, obs, args.num_agents).float().contiguous()
        obs, rew, done, info = venv.step(act)
        ret += rew[:, 0, 0]
        if (t + 1) % 100 == 0:
            print("step %5d  mean shared reward %9.3f  done envs %d" % (t + 1, float(rew[:, 0, 0].mean()), int(done[:, 0].sum())))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d envs x %d agents, %d steps: %.3g env-steps/s, %.3g agent-steps/s (policy included)"
          % (args.num_envs, total, args.steps, args.num_envs * args.steps / dt, args.num_envs * total * args.steps / dt))
