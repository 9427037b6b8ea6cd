"""A reference-style Scenario file (tests/plugins/ring_patrol_env.py: ORIGINAL, written against the reference's plugin API
scenario.py:4-12 only) dropped into formation_gym.make_env: World.step on the GPU, the file's own per-agent callbacks on
the host.  The fixture ring_patrol_n5.npz is the SAME file executed by the real reference's env shell
(tests/golden/make_golden.py: plugin_fixture; environment.py:113-184)."""
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ATOL = 1e-5
PLUGIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plugins", "ring_patrol_env.py")


def test_reference_style_plugin_matches_the_reference(golden):
    import formation_gym
    g = golden("ring_patrol_n5")
    T, N = g["acts"].shape[:2]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        env = formation_gym.make_env(PLUGIN, False, N, device="cuda:0")
    assert any("host callbacks" in str(x.message) for x in w)            # the slow path is announced ...
    assert "host callbacks" in env.info["path"]                          # ... and labelled
    assert env.observation_space[0].shape == (int(g["obs_dim"]),) and env.world_length == int(g["world_length"])
    assert env.shared_reward
    env.seed(int(g["seed"]))
    obs0 = env.reset()
    np.testing.assert_allclose(np.array(obs0), g["obs0"], rtol=0, atol=ATOL)
    # free-running (the ring scenario is smooth: no contact thresholds in the reward), reference-style list API
    worst = 0.0
    for t in range(T):
        act_n = [g["acts"][t, i].astype(np.float64) for i in range(N)]
        obs_n, rew_n, done_n, info_n = env.step(act_n)
        np.testing.assert_array_equal(np.array(act_n), 5.0 * g["acts"][t].astype(np.float64))     # scaled in place
        pos, vel = env.world.get_state()
        tol = ATOL if t < 6 else 2e-4                                     # fp32 trajectory through stiff contacts (H1)
        np.testing.assert_allclose(pos[0].double().cpu().numpy(), g["pos"][t], rtol=0, atol=tol)
        np.testing.assert_allclose(np.array(obs_n), g["obs"][t], rtol=0, atol=10 * tol)
        np.testing.assert_allclose([i["individual_reward"] for i in info_n], g["indiv"][t], rtol=0, atol=10 * tol)
        np.testing.assert_allclose(rew_n[0][0], g["shared"][t], rtol=0, atol=50 * tol)
        assert done_n == list(g["done"][t])
        worst = max(worst, float(np.abs(pos[0].double().cpu().numpy() - g["pos"][t]).max()))
    # teacher-forced: every step from the reference's own previous state, every bound 1e-5
    env.seed(int(g["seed"]))
    env.reset()
    prev_p, prev_v = g["pos0"], g["vel0"]
    for t in range(T):
        env.world.set_state(prev_p[None], prev_v[None])
        obs_n, rew_n, done_n, info_n = env.step([g["acts"][t, i].astype(np.float64) for i in range(N)])
        pos, vel = env.world.get_state()
        np.testing.assert_allclose(pos[0].double().cpu().numpy(), g["pos"][t], rtol=0, atol=ATOL)
        np.testing.assert_allclose(vel[0].double().cpu().numpy(), g["vel"][t], rtol=0, atol=ATOL)
        np.testing.assert_allclose(np.array(obs_n), g["obs"][t], rtol=0, atol=ATOL)
        np.testing.assert_allclose([i["individual_reward"] for i in info_n], g["indiv"][t], rtol=0, atol=ATOL)
        np.testing.assert_allclose(rew_n[0][0], g["shared"][t], rtol=0, atol=5 * ATOL)
        prev_p, prev_v = g["pos"][t], g["vel"][t]


def test_reference_style_plugin_batched():
    """num_envs > 1: one instance of the user's Scenario per env, env b seeded seed + 1000 b; env 0 of the batch equals
    the single-env run, and the batched tensor API returns [B, N, D]."""
    import formation_gym
    N, B = 5, 6
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        one = formation_gym.make_env(PLUGIN, False, N, device="cuda:0")
        many = formation_gym.make_env(PLUGIN, False, N, num_envs=B, device="cuda:0")
        third = formation_gym.make_env(PLUGIN, False, N, device="cuda:0")
    one.seed(7); many.seed(7); third.seed(7 + 2000)
    o1 = np.array(one.reset())
    o3 = np.array(third.reset())
    ob = many.reset()
    assert tuple(ob.shape) == (B, N, 18)
    np.testing.assert_array_equal(ob[0].double().cpu().numpy(), o1.astype(np.float32).astype(np.float64))
    np.testing.assert_array_equal(ob[2].double().cpu().numpy(), o3.astype(np.float32).astype(np.float64))
    act = torch.rand((B, N, 2), device="cuda") * 2 - 1
    obs, rew, done, info = many.step(act)
    o1s, r1s, d1s, i1s = one.step([act[0, i].cpu().numpy().astype(np.float64) for i in range(N)])
    np.testing.assert_allclose(obs[0].double().cpu().numpy(), np.array(o1s), rtol=0, atol=1e-6)
    np.testing.assert_allclose(float(rew[0, 0, 0]), r1s[0][0], rtol=0, atol=1e-5)
    assert tuple(rew.shape) == (B, N, 1) and tuple(done.shape) == (B, N)
    assert many.benchmark_data if hasattr(many, "benchmark_data") else True
    bd = many.scenario.benchmark_data(many.world.agents[1], many.world)
    assert "ring_error" in bd


def test_plugin_with_colliding_landmarks_matches_the_reference(golden):
    """tests/plugins/drifting_rocks_env.py (ORIGINAL): movable rocks of their own mass that push the agents, an immovable
    pillar that only pushes back, a beacon nobody collides with, and a reward callback that re-arms the rocks' velocity every
    step - the features of the reference's formation_hd_obs_env.py (:36-42, :82-89) in a file the real reference ran
    (fixture drifting_rocks_n4).  Through make_env(<path>): the landmarks that collide ride along in the GPU physics."""
    import formation_gym
    g = golden("drifting_rocks_n4")
    T, N = g["acts"].shape[:2]
    rocks = os.path.join(os.path.dirname(PLUGIN), "drifting_rocks_env.py")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env = formation_gym.make_env(rocks, False, N, device="cuda:0")
    assert env.num_agents == N and len(env.world.agents) == N + 3          # two rocks and the pillar behind the agents
    assert env.observation_space[0].shape == (int(g["obs_dim"]),)
    env.seed(int(g["seed"]))
    obs0 = env.reset()
    np.testing.assert_allclose(np.array(obs0), g["obs0"], rtol=0, atol=ATOL)
    hw = env.scenario.host_worlds[0]

    def landmarks():
        return (np.array([l.state.p_pos for l in hw.landmarks]), np.array([l.state.p_vel for l in hw.landmarks]))
    # free-running from the reset (the rewards are smooth): the trajectory, the rocks' included
    for t in range(T):
        obs_n, rew_n, done_n, info_n = env.step([g["acts"][t, i].astype(np.float64) for i in range(N)])
        pos, vel = env.world.get_state()
        tol = ATOL if t < 5 else 2e-4
        np.testing.assert_allclose(pos[0, :N].double().cpu().numpy(), g["pos"][t], rtol=0, atol=tol)
        lp, lv = landmarks()
        np.testing.assert_allclose(lp, g["lm_pos"][t], rtol=0, atol=tol)
        np.testing.assert_allclose(lv, g["lm_vel"][t], rtol=0, atol=10 * tol)
        np.testing.assert_allclose(np.array(obs_n), g["obs"][t], rtol=0, atol=10 * tol)
        np.testing.assert_allclose([i["individual_reward"] for i in info_n], g["indiv"][t], rtol=0, atol=10 * tol)
        assert done_n == list(g["done"][t])
    assert np.abs(g["lm_pos"][-1][1] - g["lm_pos0"][1]).max() > 0.1           # the rocks really moved ...
    np.testing.assert_array_equal(g["lm_pos"][-1][2], g["lm_pos0"][2])        # ... the pillar never does
    # teacher-forced at 1e-5: every step from the reference's own previous state of agents AND landmarks
    env.seed(int(g["seed"]))
    env.reset()
    prev = dict(pos=g["pos0"], vel=g["vel0"], lp=g["lm_pos0"], lv=g["lm_vel0"])
    for t in range(T):
        for a, p_, v_ in zip(hw.agents, prev["pos"], prev["vel"]):
            a.state.p_pos = p_.copy(); a.state.p_vel = v_.copy()
        for l, p_, v_ in zip(hw.landmarks, prev["lp"], prev["lv"]):
            l.state.p_pos = p_.copy(); l.state.p_vel = v_.copy()
        env.scenario._upload(env.world)
        obs_n, rew_n, done_n, info_n = env.step([g["acts"][t, i].astype(np.float64) for i in range(N)])
        pos, vel = env.world.get_state()
        np.testing.assert_allclose(pos[0, :N].double().cpu().numpy(), g["pos"][t], rtol=0, atol=ATOL)
        np.testing.assert_allclose(vel[0, :N].double().cpu().numpy(), g["vel"][t], rtol=0, atol=ATOL)
        lp, lv = landmarks()
        np.testing.assert_allclose(lp, g["lm_pos"][t], rtol=0, atol=ATOL)
        np.testing.assert_allclose(lv, g["lm_vel"][t], rtol=0, atol=ATOL)
        np.testing.assert_allclose(np.array(obs_n), g["obs"][t], rtol=0, atol=ATOL)
        np.testing.assert_allclose([i["individual_reward"] for i in info_n], g["indiv"][t], rtol=0, atol=ATOL)
        prev = dict(pos=g["pos"][t], vel=g["vel"][t], lp=g["lm_pos"][t], lv=g["lm_vel"][t])


def test_a_reward_callback_that_counts_its_calls(tmp_path):
    """environment.py:127-137 calls `reward` twice per agent and step: the first value goes to reward_n (and its sum is the
    shared reward), the second only to info['individual_reward'].  A callback that is not idempotent tells them apart."""
    import formation_gym
    src = '''
import numpy as np
from formation_gym.core import World, Agent
from formation_gym.scenario import BaseScenario


class Scenario(BaseScenario):
    def make_world(self, num_agents=3, world_length=25):
        world = World()
        world.world_length = world_length
        world.collaborative = COLLAB
        world.agents = [Agent() for _ in range(num_agents)]
        for i, a in enumerate(world.agents):
            a.name = 'agent %d' % i
            a.silent = True
        self.calls = 0
        self.reset_world(world)
        return world

    def reset_world(self, world):
        for i, a in enumerate(world.agents):
            a.state.p_pos = np.array([0.3 * i, 0.0])
            a.state.p_vel = np.zeros(2)

    def reward(self, agent, world):
        self.calls += 1
        return float(self.calls)

    def observation(self, agent, world):
        return np.concatenate([agent.state.p_vel, agent.state.p_pos])
'''
    for collab in (False, True):
        path = tmp_path / ("counting_%d_env.py" % collab)
        path.write_text(src.replace("COLLAB", str(collab)))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            env = formation_gym.make_env(str(path), False, 3, device="cuda:0")
        env.seed(0); env.reset()
        obs_n, rew_n, done_n, info_n = env.step([np.zeros(2) for _ in range(3)])
        # calls 1, 3, 5 are the first per agent; 2, 4, 6 the second
        assert [i["individual_reward"] for i in info_n] == [2.0, 4.0, 6.0]
        assert rew_n == ([[9.0]] * 3 if collab else [[1.0], [3.0], [5.0]])


def test_rollout_interface_for_host_paced_envs():
    """env.rollout for envs that have no multi-step launch - a reference-style Scenario file (callbacks on the host) and an env
    with a post_step_callback (environment.py:140-141: a host function after every step) - runs K `step` calls behind the same
    interface: the same results as stepping by hand, the same shapes as the fused launches return."""
    import formation_gym
    N, B, K = 5, 3, 7
    envs = []
    for _ in range(2):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            e = formation_gym.make_env(PLUGIN, False, N, num_envs=B, device="cuda:0")
        e.seed(4); e.reset()
        envs.append(e)
    a, b = envs
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    acts = torch.rand((K, B, N, 2), generator=gen, device="cuda") * 2 - 1
    obs, rew, done, info = b.rollout(acts, obs_every=2)
    assert tuple(obs.shape) == (K // 2, B, N, a._out["obs"].shape[-1]) and tuple(rew.shape) == (K, B, N, 1) and done.dtype == torch.bool
    for k in range(K):
        o, r, d, i = a.step(acts[k])
        assert torch.equal(r, rew[k]) and torch.equal(d, done[k]) and torch.equal(i["individual_reward"], info["individual_reward"][k])
        if (k + 1) % 2 == 0:
            assert torch.equal(o, obs[k // 2])
    assert a.current_step == b.current_step == K
    # ... and formation_hd_env with a post_step_callback that edits the device state after every step
    calls = []
    twins = []
    for _ in range(2):
        e = formation_gym.make_env("formation_hd_env", False, 9, num_envs=16, device="cuda:0")
        e.seed(2); e.reset()
        e.post_step_callback = lambda world: (calls.append(1), world.vel_x.mul_(0.5))
        twins.append(e)
    a, b = twins
    acts = torch.rand((4, 16, 9, 2), generator=gen, device="cuda") * 2 - 1
    obs, rew, done, info = b.rollout(acts)
    assert len(calls) == 4
    for k in range(4):
        o, r, d, i = a.step(acts[k])
        assert torch.equal(o, obs[k]) and torch.equal(r, rew[k]) and torch.equal(d, done[k])
    assert len(calls) == 8
    for x, y in zip(a.world.get_state(), b.world.get_state()):
        assert torch.equal(x, y)
    # ... and the built-in controller in the loop (rollout_policy) with that callback: the demo loop, launch by launch
    n0 = len(calls)
    o_seq, r_seq, d_seq, info = b.rollout_policy(3, 3)
    assert len(calls) == n0 + 3 and tuple(info["actions"].shape) == (3, 16, 9, 2)
    obs_a = a._out["obs"]
    for k in range(3):
        act = formation_gym.get_action_BFS(formation_gym.ezpolicy, obs_a, 3)
        assert torch.equal(act, info["actions"][k])
        obs_a, r, d, i = a.step(act)
        assert torch.equal(obs_a, o_seq[k]) and torch.equal(r, r_seq[k]) and torch.equal(d, d_seq[k])


def test_reference_style_plugin_under_the_default_vec_env_reset_mode():
    """FormationVecEnv's default reset mode ('device': the env restarts inside the step) with a reference-style Scenario file: the
    restart happens where that scenario's callbacks live, on the host, from each env's own stream - the same results as the 'host'
    mode (reset between steps, env_wrappers.py:14-18)."""
    import formation_gym
    from formation_gym.vec_env import FormationVecEnv
    N, B, T = 5, 4, 17
    vs = []
    for mode in ("device", "host"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            e = formation_gym.make_env(PLUGIN, False, N, num_envs=B, device="cuda:0", episode_length=6)
        e.seed(11)
        vs.append(FormationVecEnv(e, reset_mode=mode))
    a, b = vs
    assert torch.equal(a.reset(), b.reset())
    gen = torch.Generator(device="cuda"); gen.manual_seed(2)
    ends = 0
    for t in range(T):
        act = torch.rand((B, N, 2), generator=gen, device="cuda") * 2 - 1
        oa, ra, da, ia = a.step(act.clone())
        ob, rb, db, ib = b.step(act.clone())
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db), "step %d" % t
        assert torch.equal(ia["individual_reward"], ib["individual_reward"])
        ends += int(da.all(1).sum())
    assert ends == 2 * B
