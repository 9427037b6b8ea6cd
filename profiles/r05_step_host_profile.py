"""Where the host time of env.step goes (9 agents x 4096 envs: the kernel takes 7.7 us, the call 9.3): cProfile over 20000 calls
with a fresh action tensor object per call (what a policy produces) and with ONE re-used action tensor."""
import cProfile, pstats, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch, formation_gym
env = formation_gym.make_env("formation_hd_env", False, 9, num_envs=4096, device="cuda:0")
env.reset()
acts = [(torch.rand((4096, 9, 2), device="cuda") * 2 - 1) for _ in range(32)]
for a in acts: env.step(a)
torch.cuda.synchronize()
def loop(n, reuse):
    for k in range(n):
        env.step(acts[0] if reuse else acts[k & 31])
for reuse in (False, True):
    torch.cuda.synchronize(); t0 = time.perf_counter(); loop(20000, reuse); torch.cuda.synchronize()
    print("%s: %.2f us per env.step call (wall)" % ("one action tensor re-used" if reuse else "32 action tensors in turn", (time.perf_counter() - t0) / 20000 * 1e6))
pr = cProfile.Profile(); pr.enable(); loop(20000, False); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
