"""GPU parity of the built-in demo controller: `fg_policy_bfs` - ezpolicy expanded by
get_action_BFS, formation_gym/__init__.py:19-47 / :49-99 - against the actions the reference produced
(tests/golden/policy_n{3,9,27,81}.npz) and against the fp64 oracle.

Tolerance: 1e-5 abs (north_star).  The controller takes discrete decisions (an argsort, argmins, a 0.01 threshold);
an fp32 evaluation may legitimately decide differently only where the fp64 reference's own comparison gap is below
1e-6 (`oracle.bfs_margins`, the rule of `_check_indices` in test_gpu_parity.py), and only such rows are excused.
"""
import numpy as np
import pytest
import torch

from oracle import formation_oracle as O

pytestmark = pytest.mark.gpu

ATOL = 1e-5
NEAR_TIE = 1e-6


def _np(t):
    return t.detach().double().cpu().numpy()


def _check_actions(got, want, margins, what, scale_with_magnitude=False):
    """got / want [..., N, 2]; margins [..., N]: rows off by more than ATOL must sit on a near-tie.
    scale_with_magnitude: deep hierarchies multiply every level's output by its level number and hand 0.3 x that
    down (__init__.py:43-46, :78-79), so actions reach magnitudes of 10-40 at 10 levels (fp32 ulp 4e-6) and a
    rounding error at the top arrives amplified by prod(0.3 lev) ~ 7 at the leaves: the bound is then 1e-5 of the
    env's largest action component (>= 1), i.e. still 1e-5 abs wherever actions are O(1) as in every fixture."""
    tol = ATOL * (np.maximum(1.0, np.abs(want).max((-1, -2), keepdims=True)[..., 0]) if scale_with_magnitude else 1.0)
    bad = np.abs(got - want).max(-1) > tol
    if bad.any():
        assert (margins[bad] < NEAR_TIE).all(), "%s: %d action(s) differ away from a near-tie (max err %.3g)" % (
            what, int((bad & (margins >= NEAR_TIE)).sum()), np.abs(got - want)[bad & (margins >= NEAR_TIE)].max())
    return int(bad.sum())


def _fixture_obs(g):
    """fp64 observations the reference's policy saw at every recorded step: [T, N, 6N]."""
    T = g["act"].shape[0]
    pos = np.stack([g["pos0"]] + [g["pos"][t] for t in range(T - 1)])
    vel = np.stack([g["vel0"]] + [g["vel"][t] for t in range(T - 1)])
    shape = np.broadcast_to(g["ideal_shape"], pos.shape)
    ivel = np.broadcast_to(g["ideal_vel"], (T, 2))
    return O.observation_hd(pos, vel, shape, ivel), pos, vel


@pytest.mark.parametrize("name", ["policy_n3", "policy_n9", "policy_n27", "policy_n81",
                                  "policy_n8_per2", "policy_n16_per4", "policy_n5_per5"])
def test_policy_kernel_matches_reference_actions(golden, name):
    import formation_gym
    from formation_gym.policy_bfs import bfs_actions as _bfs
    g = golden(name)
    per = int(g["per"]) if "per" in g else 3
    bfs_actions = lambda o, p_=None, out=None: _bfs(o, per, out=out)          # the fixture's own hierarchy
    T, N = g["act"].shape[:2]
    obs64, pos, vel = _fixture_obs(g)
    margins = np.stack([O.bfs_margins(list(obs64[t]), per) for t in range(T)])
    # (1) the reference's observations rounded to fp32, every recorded step as one env of a batch of T
    obs = torch.as_tensor(obs64.astype(np.float32)).cuda()
    act = formation_gym.get_action_BFS(formation_gym.ezpolicy, obs, per)        # product entry point -> fg_policy_bfs
    assert act.shape == (T, N, 2) and act.dtype == torch.float32 and act.is_cuda
    excused = _check_actions(_np(act), g["act"], margins, name)
    assert excused <= max(1, T * N // 50)                                         # near-ties are rare
    # (2) observations written by the env's own kernel from the same states (what a rollout loop feeds back)
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=T, device="cuda:0")
    env.world.set_state(pos, vel)
    env.scenario.set_formation(env.world, np.broadcast_to(g["ideal_shape"], pos.shape), np.broadcast_to(g["ideal_vel"], (T, 2)))
    out = {"obs": env._out["obs"], "reward": env._out["reward"]}
    env.scenario.observe_batch(env.world, out)
    act2 = bfs_actions(out["obs"], 3)
    _check_actions(_np(act2), g["act"], margins, name + " (env observations)")
    # (3) a strided view (padded env pitch) and a caller-owned output give the same bits
    pad = torch.zeros((T, N * 6 * N + 10), device="cuda")
    pad[:, :N * 6 * N] = obs.reshape(T, -1)
    view = pad[:, :N * 6 * N].view(T, N, 6 * N)
    assert view.stride(0) == N * 6 * N + 10
    mine = torch.empty((T, N, 2), device="cuda")
    assert bfs_actions(view, 3, out=mine) is mine and torch.equal(mine, act)
    # (4) batch independence: a single env alone
    assert torch.equal(bfs_actions(obs[T // 2:T // 2 + 1].contiguous(), 3)[0], act[T // 2])


@pytest.mark.parametrize("N,per,B", [(8, 2, 40), (16, 4, 33), (25, 5, 20), (64, 8, 9), (243, 3, 5), (729, 3, 2), (1024, 2, 2),
                                     (5, 5, 64), (7, 7, 10), (125, 5, 3)])
def test_policy_kernel_other_hierarchies_against_oracle(N, per, B):
    """Any N = per^L, 2 <= per <= 8 (one level, deep binary trees, the largest agent counts): fp64 oracle on
    the same fp32 observations."""
    from formation_gym.policy_bfs import bfs_actions
    st = O.reset_hd(7 + 1000 * np.arange(B), N)
    st["pos"] *= 0.5
    f32 = lambda x: np.asarray(x, dtype=np.float32).astype(np.float64)
    obs64 = O.observation_hd(f32(st["pos"]), st["vel"], f32(st["ideal_shape"]), f32(st["ideal_vel"]))
    obs = torch.as_tensor(obs64.astype(np.float32)).cuda()
    act = _np(bfs_actions(obs, per))
    nb = B if N <= 64 else 2                                   # the Python queue walk is slow at large N
    want = np.stack([np.array(O.get_action_bfs(O.ezpolicy, list(_np(obs[b])), per, strict=False)) for b in range(nb)])
    margins = np.stack([O.bfs_margins(list(_np(obs[b])), per) for b in range(nb)])
    _check_actions(act[:nb], want, margins, "N=%d per=%d" % (N, per), scale_with_magnitude=True)
    assert np.isfinite(act).all()


def test_policy_errors_and_empty_batch():
    from formation_gym import _native
    from formation_gym.policy_bfs import bfs_actions
    assert bfs_actions(torch.zeros((0, 9, 54), device="cuda"), 3).shape == (0, 9, 2)
    with pytest.raises(_native.FormationHipError) as e:
        bfs_actions(torch.zeros((2, 10, 60), device="cuda"), 3)
    assert e.value.code == _native.FG_ERR_UNSUPPORTED_N
    with pytest.raises(ValueError):
        bfs_actions(torch.zeros((2, 9, 50), device="cuda"), 3)
    with pytest.raises(ValueError):
        bfs_actions(torch.zeros((2, 9, 54)), 3)                 # the HIP path takes device tensors only


def test_policy_closed_loop_reduces_formation_error_and_tracks_reference(golden):
    """Closed loop through env.step with the HIP controller: (a) free-running from the fixture's initial state it
    follows the reference's own closed-loop trajectory over a short horizon (the loop has no contacts at this
    density, so fp32 rounding does not blow up); (b) over an episode the shared reward improves in > 90 % of envs."""
    import formation_gym
    g = golden("policy_n9")
    N = 9
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=1, device="cuda:0")
    env.world.set_state(g["pos0"][None], g["vel0"][None])
    env.scenario.set_formation(env.world, g["ideal_shape"][None], g["ideal_vel"][None])
    env.world.step_count.zero_()
    obs = env._out["obs"]
    env.scenario.observe_batch(env.world, {"obs": obs, "reward": env._out["reward"]})
    obs64, _, _ = _fixture_obs(g)
    for t in range(10):
        if O.bfs_margins(list(obs64[t]), 3).min() < 1e-4:      # a decision the accumulated fp32 drift could flip
            break
        act = formation_gym.get_action_BFS(formation_gym.ezpolicy, obs, 3)
        np.testing.assert_allclose(_np(act)[0], g["act"][t], rtol=0, atol=1e-4)
        obs, rew, done, info = env.step(act)
        pos, _ = env.world.get_state()
        np.testing.assert_allclose(_np(pos)[0], g["pos"][t], rtol=0, atol=1e-4)
    B = 256                                                   # at N = 9 the controller improves the reward
    env = formation_gym.make_env("formation_hd_env", False, 9, num_envs=B, device="cuda:0")
    env.seed(5)
    obs = env.reset()
    first = None
    for t in range(60):
        obs, rew, done, info = env.step(formation_gym.get_action_BFS(formation_gym.ezpolicy, obs, 3))
        if first is None:
            first = rew[:, 0, 0].clone()
    assert (rew[:, 0, 0] > first).float().mean() > 0.9


@pytest.mark.parametrize("N,B,K,per", [(27, 50, 12, 3), (9, 70, 9, 3), (3, 33, 6, 3), (81, 6, 5, 3), (243, 3, 4, 3),
                                        (27, 4096, 20, 3), (16, 20, 5, 4), (8, 30, 6, 2), (25, 7, 4, 5),
                                        # the other hierarchies' closed loops, now inside the pipelined kernels (one launch)
                                        (4, 40, 5, 2), (4, 40, 5, 4), (8, 30, 5, 8), (16, 33, 5, 2), (32, 9, 4, 2), (64, 5, 4, 2),
                                        (64, 5, 4, 4), (64, 5, 4, 8), (125, 3, 3, 5), (16, 8192, 4, 4), (64, 2048, 3, 4), (25, 3000, 4, 5),
                                        (9, 5000, 4, 3), (9, 8200, 3, 3),      # 9 agents, closed loop with the gather writer
                                        (3, 32800, 4, 3), (4, 32768, 4, 2), (4, 33000, 3, 4),   # one env per lane, controller on registers
                                        (3, 98400, 20, 3), (4, 98354, 12, 2),                   # ... two producer waves per workgroup
                                        (8, 5000, 3, 8), (8, 65600, 2, 8),                       # 8 agents: gather writer, two / one writer waves
                                        (8, 36870, 3, 2), (8, 12300, 3, 8),                      # ... and 32-env workgroups from 12288 envs
                                        (36, 6, 3, 6)])          # 6^2: no pipelined instantiation - chained launches
def test_closed_loop_rollout_equals_policy_plus_step_calls(N, B, K, per):
    """env.rollout_policy(K) - the controller inside the pipelined rollout kernels (N = per^L up to 243 agents for per 2, 3,
    4, 5, 8) or chained launches (other hierarchies) - equals K x (get_action_BFS on the last observation, env.step) bit for bit, device
    auto-reset at mixed episode phases included; the state-based controller launch gives the same actions."""
    import formation_gym
    envs = []
    step0 = np.where(np.arange(B) % 3 == 0, 100 - 2 - (np.arange(B) // 3) % max(K - 1, 1), 5)
    for _ in range(2):
        e = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
        e.scenario.seed(17)
        e.scenario.reset_device(e.world, rng_offset=777)
        e.world.pos_x.mul_(0.5); e.world.pos_y.mul_(0.5)
        e.world.step_count.copy_(torch.as_tensor(step0, dtype=torch.int32))
        e.auto_reset = True
        envs.append(e)
    a, b = envs
    obs_seq, rew_seq, done_seq, info_seq = b.rollout_policy(K, per)
    assert info_seq["actions"].shape == (K, B, N, 2)
    obs = a._out["obs"]
    a.scenario.observe_batch(a.world, {"obs": obs, "reward": a._out["reward"]})
    for k in range(K):
        act = formation_gym.get_action_BFS(formation_gym.ezpolicy, obs, per)
        assert torch.equal(act, a.scenario.policy_actions(a.world, per))        # from the state: same bits
        assert torch.equal(act, info_seq["actions"][k]), "actions differ at step %d" % k
        obs, rew, done, info = a.step(act)
        assert torch.equal(obs, obs_seq[k]) and torch.equal(rew, rew_seq[k]) and torch.equal(done, done_seq[k])
        assert torch.equal(info["individual_reward"], info_seq["individual_reward"][k])
    assert done_seq.any() and not done_seq.all()
    for x, y in zip(a.world.get_state(), b.world.get_state()):
        assert torch.equal(x, y)
    assert torch.equal(a.world.step_count, b.world.step_count) and torch.equal(a.scenario.ideal_shape, b.scenario.ideal_shape)
    # caller-owned buffers (launch bound once per buffer set), one step per launch, continuing the episode
    f = dict(dtype=torch.float32, device="cuda")
    bufs = dict(obs=torch.empty((1, B, N, 6 * N), **f), reward=torch.empty((1, B, N), **f), indiv=torch.empty((1, B, N), **f),
                done=torch.zeros((1, B, N), dtype=torch.uint8, device="cuda"), act=torch.empty((1, B, N, 2), **f))
    for _ in range(3):
        o1, r1, d1, i1 = b.rollout_policy(1, per, out=bufs)
        act = formation_gym.get_action_BFS(formation_gym.ezpolicy, obs, per)
        obs, rew, done, info = a.step(act)
        assert torch.equal(i1["actions"][0], act) and torch.equal(o1[0], obs) and torch.equal(r1[0], rew)
    assert len(b._roll_launchers) <= 2                     # this buffer set + the env's own default buffers
    # obs_every: rewards unchanged, every 2nd observation
    c = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
    c.scenario.seed(17); c.scenario.reset_device(c.world, rng_offset=777)
    c.world.pos_x.mul_(0.5); c.world.pos_y.mul_(0.5)
    c.world.step_count.copy_(torch.as_tensor(step0, dtype=torch.int32)); c.auto_reset = True
    o2, r2, _, _ = c.rollout_policy(K, per, obs_every=2)
    assert torch.equal(r2, rew_seq)
    for s_ in range(K // 2):
        assert torch.equal(o2[s_], obs_seq[2 * s_ + 1])
