"""Randomised differential test of the K-step launches: for seeded random draws of (agent count, batch size, steps per
launch, obs_every, padded observation pitch, episode phases, controller on / off) `env.rollout` / `env.rollout_policy`
must equal the same number of `env.step` calls bit for bit - observations (also into padded buffers), rewards, dones,
reset draws, final state (the K-step side at times with its RNG offset in device memory).  Batch sizes straddle the workgroup env counts (4, 8, 16) and the host's thresholds (the
27-agent HBM-streaming tile writer needs a rollout buffer beyond the Infinity Cache; the 9-agent variants switch at 4096
and 8192 envs), so every instantiation and its edge handling (partial last workgroup, step slots off the 128-byte grid)
is visited."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SEEDS = range(int(os.environ.get("FG_FUZZ_SEEDS", "36")))      # a soak run: FG_FUZZ_SEEDS=400 pytest -m gpu tests/test_gpu_fuzz_rollout.py


def _case(rs):
    N = int(rs.choice([3, 9, 27, 27, 27, 81, 243, 10, 100, 300]))
    if N == 27:
        B = int(rs.choice([1, 15, 16, 17, 33, 1300 + rs.randint(0, 40), 2048 + rs.randint(-3, 4), 16384 + rs.randint(-2, 3)]))
    elif N == 9:
        B = int(rs.choice([1, 5, 64, 4096 + rs.randint(-2, 3), 8192 + rs.randint(-2, 3), 12001]))
    elif N == 3:
        B = int(rs.choice([1, 17, 1000]))
    elif N == 81:
        B = int(rs.choice([1, 3, 4, 5, 128, 129, 130]))      # the split single step (<= 128 envs) against the pipelined rollout
    elif N == 243:
        B = int(rs.choice([1, 2, 5, 9, 96, 97]))
    elif N >= 100:
        B = int(rs.choice([1, 3, 40]))                        # run-time N, one env per workgroup: split single steps, K-loop launches
    else:
        B = int(rs.choice([1, 7, 50]))
    big = N == 27 and 1300 <= B < 16384
    K = int(rs.choice([20, 21, 24]) if big else rs.randint(1, 9))
    every = int(rs.choice([1, 1, 1, 2, 3]))
    pitch_kind = int(rs.choice([0, 0, 1, 2]))
    policy = bool(N in (3, 9, 27, 81, 243) and rs.rand() < 0.35)
    if (N == 27 and B > 4096) or (N == 243 and B > 9):
        K = min(K, 4)
    counter = bool(rs.rand() < 0.3)                           # the K-step side keeps its RNG offset in device memory
    return N, B, K, every, pitch_kind, policy, counter


@pytest.mark.parametrize("seed", SEEDS)
def test_random_rollout_equals_step_calls(seed):
    import formation_gym
    rs = np.random.RandomState(1000 + seed)
    N, B, K, every, pitch_kind, policy, counter = _case(rs)
    envs = []
    step0 = rs.randint(0, 100, B)
    step0[rs.rand(B) < 0.3] = 100 - 1 - rs.randint(0, max(K, 1))            # some episodes end inside the launch
    for _ in range(2):
        e = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
        e.scenario.seed(seed)
        e.scenario.reset_device(e.world, rng_offset=seed)
        e.world.pos_x.mul_(0.5); e.world.pos_y.mul_(0.5)
        e.world.step_count.copy_(torch.as_tensor(step0, dtype=torch.int32))
        e.auto_reset = True
        envs.append(e)
    a, b = envs
    if counter:
        b.use_device_rng_counter()
    D = 6 * N
    pitch = {0: N * D, 1: -(-N * D // 32) * 32, 2: -(-N * D // 32) * 32 + 64}[pitch_kind]
    f = dict(dtype=torch.float32, device="cuda")
    slots = K // every
    buf = torch.full((max(slots, 1), B, pitch), -3.0, **f)
    out = dict(obs=buf[:slots, :, :N * D].view(slots, B, N, D), reward=torch.empty((K, B, N), **f),
               indiv=torch.empty((K, B, N), **f), done=torch.zeros((K, B, N), dtype=torch.uint8, device="cuda"))
    acts = torch.as_tensor(rs.uniform(-1, 1, (K, B, N, 2)).astype(np.float32)).cuda()
    tag = "seed %d: N=%d B=%d K=%d every=%d pitch=%d policy=%s counter=%s" % (seed, N, B, K, every, pitch, policy, counter)
    if policy:
        out["act"] = torch.empty((K, B, N, 2), **f)
        obs_seq, rew_seq, done_seq, info_seq = b.rollout_policy(K, 3, out=out, obs_every=every)
        obs = a._out["obs"]
        a.scenario.observe_batch(a.world, {"obs": obs, "reward": a._out["reward"]})
    else:
        obs_seq, rew_seq, done_seq, info_seq = b.rollout(acts, out=out, obs_every=every)
    for k in range(K):
        act = formation_gym.get_action_BFS(formation_gym.ezpolicy, obs, 3) if policy else acts[k]
        if policy:
            assert torch.equal(act, info_seq["actions"][k]), tag + " actions, step %d" % k
        obs, rew, done, info = a.step(act)
        if (k + 1) % every == 0:
            assert torch.equal(obs, obs_seq[k // every]), tag + " observations, step %d" % k
        assert torch.equal(rew, rew_seq[k]) and torch.equal(done, done_seq[k]), tag + " reward / done, step %d" % k
        assert torch.equal(info["individual_reward"], info_seq["individual_reward"][k]), tag
    for x, y in zip(a.world.get_state(), b.world.get_state()):
        assert torch.equal(x, y), tag + " final state"
    assert torch.equal(a.world.step_count, b.world.step_count) and torch.equal(a.scenario.ideal_shape, b.scenario.ideal_shape), tag
    pad = buf[:slots, :, N * D:]
    assert ((pad == -3.0) | (pad == 0.0)).all(), tag + " pad"
