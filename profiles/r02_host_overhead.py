"""Host-side cost of the per-step API calls (async launches of a tiny batch, so the host is the bound): env.step 7.7 us per call (4.3 of them the
ctypes call), get_action_BFS on a device tensor 7.0 us (9.5 before the stream handle and the entry point were cached); cProfile listing of both."""
import sys, time, cProfile, pstats
sys.path[:0] = ["/root/repo", "/root/repo/gym-formation_amd"]
import torch, formation_gym
env = formation_gym.make_env("formation_hd_env", False, 3, num_envs=64, device="cuda:0")
env.seed(1); env.reset(); env.auto_reset = True
act = torch.zeros((64, 3, 2), device="cuda")
for _ in range(200): env.step(act)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5000): env.step(act)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("env.step host us/call: %.2f" % ((t1 - t) / 5000 * 1e6))
obs = env._out["obs"]
t = time.perf_counter()
for _ in range(5000): a = formation_gym.get_action_BFS(formation_gym.ezpolicy, obs, 3)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("get_action_BFS host us/call: %.2f" % ((t1 - t) / 5000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(3000): env.step(act)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
pr = cProfile.Profile(); pr.enable()
for _ in range(3000): a = formation_gym.get_action_BFS(formation_gym.ezpolicy, obs, 3)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
