"""GPU parity of the World features beyond the reference's own scenarios:
agents of different mass / size / accel / max_speed (core.py:68-75, 97-99, 289-322) and non-silent agents
(core.py:279-286, formation_hd_env.py:48-51).  Fixtures come from the real reference (tests/golden/make_golden.py),
every fp32 bound is 1e-5 abs (shared reward: 2e-6 relative, H2)."""
import numpy as np
import pytest
import torch

import formation_gym
from oracle import formation_oracle as O
from tests.test_gpu_parity import ATOL, _load, _make, _np

pytestmark = pytest.mark.gpu


def _apply_hetero(env, g, walls=False):
    from formation_gym.core import Wall
    for a, m, s, ac, ms in zip(env.world.agents, g["agent_mass"], g["agent_size"], g["agent_accel"], g["agent_max_speed"]):
        a.initial_mass = float(m); a.size = float(s)
        a.accel = None if np.isnan(ac) else float(ac)
        a.max_speed = None if np.isnan(ms) else float(ms)
    if walls:
        env.world.walls = [Wall(o, ax, ep, w) for (o, ax, ep, w) in O.GOLDEN_WALLS]


@pytest.mark.parametrize("name,walls", [("hd_n9_masses", False), ("hd_n27_masses", True)])
def test_per_agent_mass_size_options_teacher_forced(golden, name, walls):
    g = golden(name)
    T, B, N = g["acts"].shape[:3]
    env = _make(N, B)
    _apply_hetero(env, g, walls)
    env.enable_assignments(True)
    prev_pos, prev_vel = g["pos0"], g["vel0"]
    worst = {"pos": 0.0, "vel": 0.0, "indiv": 0.0, "obs": 0.0}
    for t in range(T):
        _load(env, prev_pos, prev_vel, g["ideal_shape"], g["ideal_vel"], np.full(B, t))
        obs, rew, done, info = env.step(torch.as_tensor(g["acts"][t]).cuda())
        pos, vel = env.world.get_state()
        np.testing.assert_allclose(_np(pos), g["pos"][t], rtol=0, atol=ATOL)
        np.testing.assert_allclose(_np(vel), g["vel"][t], rtol=0, atol=ATOL)
        ok = g["cnt_margin"][t] > 1e-5
        np.testing.assert_allclose(_np(info["individual_reward"])[ok], g["indiv"][t][ok], rtol=0, atol=ATOL)
        np.testing.assert_allclose(_np(rew)[ok, :, 0], g["shared"][t][ok], rtol=2e-6, atol=ATOL)
        np.testing.assert_array_equal(done.cpu().numpy(), g["done"][t])
        worst["pos"] = max(worst["pos"], np.abs(_np(pos) - g["pos"][t]).max())
        worst["vel"] = max(worst["vel"], np.abs(_np(vel) - g["vel"][t]).max())
        if (t + 1) in g["obs_steps"]:
            np.testing.assert_allclose(_np(obs), g["obs_t%d" % (t + 1)], rtol=0, atol=ATOL)
        prev_pos, prev_vel = g["pos"][t], g["vel"][t]
    assert g["cnt"].sum() > 0
    # a K-step launch with the same table runs step_kernel's K-loop: bit-identical to single steps
    _load(env, g["pos0"], g["vel0"], g["ideal_shape"], g["ideal_vel"], np.zeros(B))
    acts = torch.as_tensor(g["acts"][:4]).cuda().contiguous()
    singles = []
    for t in range(4):
        o, r, d, _ = env.step(acts[t])
        singles.append((o.clone(), r.clone()))
    _load(env, g["pos0"], g["vel0"], g["ideal_shape"], g["ideal_vel"], np.zeros(B))
    o_seq, r_seq, _, _ = env.rollout(acts)
    for t in range(4):
        assert torch.equal(o_seq[t], singles[t][0]) and torch.equal(r_seq[t], singles[t][1])


@pytest.mark.parametrize("N,B", [(3, 50), (12, 33), (27, 40), (70, 9), (81, 6), (200, 3)])
def test_random_per_agent_tables_against_oracle(N, B):
    """Randomised tables over specialised, run-time and whole-workgroup agent counts: one teacher-forced step vs the
    fp64 oracle on the same fp32 inputs, crowded so that contacts and penalties occur."""
    rs = np.random.RandomState(1000 + N)
    env = _make(N, B)
    mass = rs.uniform(0.4, 4.0, N); size = rs.uniform(0.01, 0.08, N)
    accel = np.where(rs.uniform(size=N) < 0.5, rs.uniform(1.0, 7.0, N), np.nan)
    vmax = np.where(rs.uniform(size=N) < 0.5, rs.uniform(0.1, 1.0, N), np.nan)
    g = dict(agent_mass=mass, agent_size=size, agent_accel=accel, agent_max_speed=vmax)
    _apply_hetero(env, g)
    f32 = lambda x: np.asarray(x, dtype=np.float32).astype(np.float64)
    spread = 0.5 * np.sqrt(N / 27.0)
    pos = f32(rs.uniform(-spread, spread, (B, N, 2))); vel = f32(rs.uniform(-0.5, 0.5, (B, N, 2)))
    shape = rs.uniform(-1, 1, (B, N, 2)); shape = f32(shape - shape.mean(1, keepdims=True))
    ivel = f32(rs.uniform(-1, 1, (B, 2)))
    act = rs.uniform(-1, 1, (B, N, 2)).astype(np.float32)
    _load(env, pos, vel, shape, ivel, np.zeros(B))
    obs, rew, done, info = env.step(torch.as_tensor(act).cuda())
    P = O.HdParams(); P.agent_size = float(np.float32(size[0]))          # the scale of collide_thresh is agent 0's
    st = dict(pos=pos, vel=vel, ideal_shape=shape, ideal_vel=ivel, step=np.zeros(B, dtype=np.int32))
    opts = dict(mass=f32(mass), size=f32(size), accel=np.where(np.isnan(accel), np.nan, f32(np.nan_to_num(accel))),
                max_speed=np.where(np.isnan(vmax), np.nan, f32(np.nan_to_num(vmax))))
    new, out = O.step_hd(st, act.astype(np.float64), **opts)
    p_, v_ = env.world.get_state()
    np.testing.assert_allclose(_np(p_), new["pos"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(_np(v_), new["vel"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(_np(obs), out["obs"], rtol=0, atol=ATOL)
    ok = out["cnt_margin"] > 1e-5
    assert ok.any() and out["cnt"].sum() > 0
    np.testing.assert_allclose(_np(info["individual_reward"])[ok], out["indiv"][ok], rtol=0, atol=ATOL)


def test_identical_table_rows_equal_the_scalar_description():
    """A table whose rows all equal the scalars must give what the scalar path gives (same formula, 1e-6: the table path
    sums the pair forces in a plain loop, the scalar one in the packed loops)."""
    N, B = 27, 64
    e1, e2 = _make(N, B), _make(N, B)
    for e in (e1, e2):
        e.seed(4); e.reset()
        e.world.set_state(pos=e.world.get_state()[0] * 0.3)
    p = e2.scenario.params(e2.world)
    rows = torch.tensor([[1.0, 0.03, 0, 0, 0, -1.0, 0, 0]] * N, dtype=torch.float32, device="cuda")
    act = torch.rand((B, N, 2), device="cuda") * 2 - 1
    o1, r1, _, i1 = e1.step(act.clone())
    from formation_gym import _native
    p.agent_props = rows.data_ptr()
    out = e2._out
    _native.check(_native.load().fg_step_hd(
        p, B, N, e2.world.pos_x.data_ptr(), e2.world.pos_y.data_ptr(), e2.world.vel_x.data_ptr(), e2.world.vel_y.data_ptr(),
        act.data_ptr(), e2.scenario.ideal_shape.data_ptr(), e2.scenario.ideal_vel.data_ptr(), e2.world.step_count.data_ptr(),
        out["obs"].data_ptr(), out["reward"].data_ptr(), out["indiv"].data_ptr(), out["done"].data_ptr(), None, None, None,
        _native.current_stream(e2.world.device)))
    torch.cuda.synchronize()
    assert float((out["obs"] - o1).abs().max()) < 1e-6 and float((out["indiv"] - i1["individual_reward"]).abs().max()) < 1e-5
    np.testing.assert_allclose(_np(e2.world.pos_x), _np(e1.world.pos_x), rtol=0, atol=1e-6)


def test_non_silent_agents_through_the_world_api(golden):
    """The fixture's path: agent.action.u / .c set per agent, world.step(), scenario.observation / reward per agent
    (the reference's env.step raises IndexError for non-silent agents, and so does this one)."""
    import formation_gym
    g = golden("hd_n5_comm")
    T, N = g["acts"].shape[:2]
    env = formation_gym.make_env("formation_hd_env", False, N, device="cuda:0")
    world, sc = env.world, env.scenario
    for a, s in zip(world.agents, g["silent"]):
        a.silent = bool(s)
    env2 = formation_gym.MultiAgentEnv(world, sc.reset_world, sc.reward, sc.observation)
    assert [type(s).__name__ for s in env2.action_space] == ["Tuple" if not s else "Box" for s in g["silent"]]
    with pytest.raises(IndexError, match="list index out of range"):
        env2.step([np.zeros(2) for _ in range(N)])
    prev_pos, prev_vel = g["pos0"], g["vel0"]
    for t in range(T):
        _load(env, prev_pos[None], prev_vel[None], g["ideal_shape"][None], g["ideal_vel"][None], np.zeros(1))   # teacher-forced
        for i, a in enumerate(world.agents):
            a.action.u = torch.as_tensor(g["acts"][t, i])[None]            # RAW action: the x5 of _set_action happens in-kernel
            a.action.c = torch.as_tensor(g["comm"][t, i])[None]
        world.step()
        pos, vel = world.get_state()
        np.testing.assert_allclose(_np(pos)[0], g["pos"][t], rtol=0, atol=ATOL)
        np.testing.assert_allclose(_np(vel)[0], g["vel"][t], rtol=0, atol=ATOL)
        c = np.stack([_np(a.state.c)[0] for a in world.agents])
        np.testing.assert_array_equal(c, g["c"][t].astype(np.float32).astype(np.float64))
        prev_pos, prev_vel = g["pos"][t], g["vel"][t]
        obs = np.stack([_np(sc.observation(a, world))[0] for a in world.agents])
        np.testing.assert_allclose(obs, g["obs"][t], rtol=0, atol=ATOL)
        rew = np.array([float(sc.reward(a, world)[0]) for a in world.agents])
        np.testing.assert_allclose(rew, g["indiv"][t], rtol=0, atol=ATOL)
    assert (g["obs"][:, 0, 2 * N:4 * N - 2] != 0).any()


@pytest.mark.parametrize("N,B", [(9, 130), (27, 70), (100, 5), (81, 40)])
def test_comm_block_at_batch_sizes_and_launch_paths(N, B):
    """fg_observe_hd / fg_step_hd with FgParams.comm_state over lane-group, whole-workgroup and split launches
    vs the oracle's observation; c_noise draws are Gaussian with the requested scale."""
    import formation_gym
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
    env.seed(5); env.reset()
    world, sc = env.world, env.scenario
    silent = np.zeros(N, dtype=bool); silent[::4] = True
    for a, s in zip(world.agents, silent):
        a.silent = bool(s)
    comm_c, action_c = world.ensure_comm()
    action_c.copy_(torch.rand((B, N, 2), device="cuda"))
    world.action_u.copy_(torch.rand((B, N, 2), device="cuda") * 2 - 1)
    st0 = [t.clone() for t in world.get_state()]
    world.step()
    want_c = O.update_comm(_np(action_c), silent)
    np.testing.assert_array_equal(_np(comm_c), want_c)
    pos, vel = world.get_state()
    out = dict(obs=torch.empty((B, N, 6 * N), device="cuda"), reward=torch.empty((B, N), device="cuda"))
    sc.observe_batch(world, out)
    want = O.observation_hd(_np(pos), _np(vel), _np(sc.ideal_shape), _np(sc.ideal_vel), comm=want_c)
    np.testing.assert_allclose(_np(out["obs"]), want, rtol=0, atol=ATOL)
    # the fused step with the communication block (fg_step_hd + comm_state; split launches at few envs)
    act = torch.rand((B, N, 2), device="cuda") * 2 - 1
    st = dict(pos=_np(pos), vel=_np(vel), ideal_shape=_np(sc.ideal_shape), ideal_vel=_np(sc.ideal_vel),
              step=world.step_count.cpu().numpy())
    sc.step_batch(world, act, env._out)
    new, ref = O.step_hd(st, _np(act), comm=want_c)
    np.testing.assert_allclose(_np(env._out["obs"]), ref["obs"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(_np(env._out["indiv"]), ref["indiv"], rtol=0, atol=ATOL)
    # noise: N(0, c_noise^2) around action.c for the non-silent agents, fresh every step
    for a in world.agents:
        a.c_noise = 0.25
    world.step()
    d1 = (_np(comm_c) - _np(action_c))[:, ~silent]
    world.step()
    d2 = (_np(comm_c) - _np(action_c))[:, ~silent]
    tol = 4.5 * 0.25 / np.sqrt(d1.size)                      # 4.5 standard errors of the mean (std: / sqrt(2), well inside)
    assert abs(d1.std() - 0.25) < tol and abs(d1.mean()) < tol and np.abs(d1 - d2).max() > 0.1
    assert (_np(comm_c)[:, silent] == 0).all()


def test_non_colliding_agents_ghosts_and_a_soft_wall_teacher_forced(golden):
    """Entity.collide / Entity.ghost per agent (core.py:54-58, 292-293, 326-327) with hard walls and a soft one, agents of
    different mass and size, through env.step: the reference's trajectory, teacher-forced at 1e-5; a K-step launch gives
    the single steps' bits."""
    from formation_gym.core import Wall
    g = golden("hd_n9_flags")
    T, B, N = g["acts"].shape[:3]
    assert (~g["agent_collide"]).any() and g["agent_ghost"].any()

    def build():
        env = _make(N, B)
        _apply_hetero(env, g, walls=True)
        env.world.walls = env.world.walls + [Wall(o, ax, ep, w, hard=False) for (o, ax, ep, w) in O.GOLDEN_SOFT_WALLS]
        for a, c, gh in zip(env.world.agents, g["agent_collide"], g["agent_ghost"]):
            a.collide = bool(c); a.ghost = bool(gh)
        return env
    env = build()
    prev_pos, prev_vel = g["pos0"], g["vel0"]
    felt_soft_wall = False
    for t in range(T):
        _load(env, prev_pos, prev_vel, g["ideal_shape"], g["ideal_vel"], np.full(B, t))
        obs, rew, done, info = env.step(torch.as_tensor(g["acts"][t]).cuda())
        pos, vel = env.world.get_state()
        np.testing.assert_allclose(_np(pos), g["pos"][t], rtol=0, atol=ATOL)
        np.testing.assert_allclose(_np(vel), g["vel"][t], rtol=0, atol=ATOL)
        ok = g["cnt_margin"][t] > 1e-5
        np.testing.assert_allclose(_np(info["individual_reward"])[ok], g["indiv"][t][ok], rtol=0, atol=ATOL)
        if (t + 1) in g["obs_steps"]:
            np.testing.assert_allclose(_np(obs), g["obs_t%d" % (t + 1)], rtol=0, atol=ATOL)
        # the fixture means something: without the flags the oracle's step differs from the reference's
        plain, _ = O.step_hd(dict(pos=prev_pos, vel=prev_vel, ideal_shape=g["ideal_shape"], ideal_vel=g["ideal_vel"], step=np.full(B, t)),
                             g["acts"][t].astype(np.float64), mass=g["agent_mass"], size=g["agent_size"], accel=g["agent_accel"],
                             max_speed=g["agent_max_speed"], walls=O.GOLDEN_WALLS + [w + (False,) for w in O.GOLDEN_SOFT_WALLS])
        felt_soft_wall = felt_soft_wall or np.abs(plain["pos"] - g["pos"][t]).max() > 1e-4
        prev_pos, prev_vel = g["pos"][t], g["vel"][t]
    assert felt_soft_wall
    # a non-colliding agent collects no collision penalties (formation_hd_env.py:71)
    e1, e2 = build(), build()
    for e in (e1, e2):
        _load(e, g["pos0"], g["vel0"], g["ideal_shape"], g["ideal_vel"], np.zeros(B))
    acts = torch.as_tensor(g["acts"][:6]).cuda().contiguous()
    o_seq, r_seq, _, i_seq = e1.rollout(acts)
    for t in range(6):
        o, r, _, i = e2.step(acts[t])
        assert torch.equal(o, o_seq[t]) and torch.equal(r, r_seq[t]) and torch.equal(i["individual_reward"], i_seq["individual_reward"][t])


def test_immovable_agent_through_the_world_api(golden):
    """An immovable agent (core.py:231, 266-267, 294-295, 319-321) among agents of different mass, one of them not
    colliding: agent.action.u set per agent, world.step(), scenario.observation / reward (the reference's env.step asserts
    on a silent immovable agent, environment.py:236, and so does this one)."""
    import formation_gym
    g = golden("hd_n6_immovable")
    T, N = g["acts"].shape[:2]
    assert str(g["env_step_raises"]).startswith("AssertionError")
    env = formation_gym.make_env("formation_hd_env", False, N, device="cuda:0")
    world, sc = env.world, env.scenario
    for a, mv, cl, m in zip(world.agents, g["movable"], g["collide"], g["mass"]):
        a.movable = bool(mv); a.collide = bool(cl); a.initial_mass = float(m)
    with pytest.raises(AssertionError):
        env.step([np.zeros(2) for _ in range(N)])
    with pytest.raises(AssertionError):
        env.step(torch.zeros((1, N, 2), device="cuda"))
    prev_pos, prev_vel = g["pos0"], g["vel0"]
    frozen = int(np.flatnonzero(~g["movable"])[0])
    for t in range(T):
        _load(env, prev_pos[None], prev_vel[None], g["ideal_shape"][None], g["ideal_vel"][None], np.zeros(1))   # teacher-forced
        for i, a in enumerate(world.agents):
            a.action.u = torch.as_tensor(g["acts"][t, i])[None]            # RAW action: the x5 of _set_action happens in-kernel
        world.step()
        pos, vel = world.get_state()
        np.testing.assert_allclose(_np(pos)[0], g["pos"][t], rtol=0, atol=ATOL)
        np.testing.assert_allclose(_np(vel)[0], g["vel"][t], rtol=0, atol=ATOL)
        np.testing.assert_array_equal(_np(pos)[0, frozen], g["pos0"][frozen].astype(np.float32).astype(np.float64))   # never moves
        obs = np.stack([_np(sc.observation(a, world))[0] for a in world.agents])
        np.testing.assert_allclose(obs, g["obs"][t], rtol=0, atol=ATOL)
        rew = np.array([float(sc.reward(a, world)[0]) for a in world.agents])
        np.testing.assert_allclose(rew, g["indiv"][t], rtol=0, atol=ATOL)
        prev_pos, prev_vel = g["pos"][t], g["vel"][t]
    assert np.abs(g["vel"][:, frozen] - g["vel0"][frozen]).max() == 0       # (the reference leaves its velocity alone too)


def test_scripted_agents_through_the_world_api(golden):
    """Scripted agents (Agent.action_callback, core.py:152-158, 210-211) through World.step: the callback is BATCHED - called
    once per step with the Agent (state.p_pos / p_vel are [B, 2] device tensors) and the World, returning the action of all B
    envs - and its `u` is used as it is (no sensitivity: FG_AGENT_SCRIPTED); the policy agents' raw actions get the x5 of
    _set_action in-kernel.  Fixture hd_n6_scripted: the same callback run by the reference's core.py, one env; here the env is
    replicated so that the batched contract is exercised.  env.step refuses (the fused launch has nowhere to call back)."""
    import formation_gym
    g = golden("hd_n6_scripted")
    T, N = g["acts"].shape[:2]
    B = 5
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
    world, sc = env.world, env.scenario
    calls = []

    def callback(agent, w):                                   # tests/golden/make_golden.py scripted_u on [B, 2] tensors
        p, v, lead = agent.state.p_pos, agent.state.p_vel, w.agents[0].state.p_pos
        calls.append(tuple(p.shape))
        act = formation_gym.core.Action()
        act.u = 0.6 * torch.stack((-p[..., 1], p[..., 0]), -1) - 0.3 * v + 0.2 * (lead - p)
        return act
    for a, s_, m in zip(world.agents, g["scripted"], g["mass"]):
        a.initial_mass = float(m)
        if s_:
            a.action_callback = callback
    assert len(world.scripted_agents) == 2 and len(world.policy_agents) == N - 2
    with pytest.raises(NotImplementedError):
        env.step(torch.zeros((B, N, 2), device="cuda"))
    rep = lambda x: np.repeat(np.asarray(x)[None], B, 0)
    prev_pos, prev_vel = g["pos0"], g["vel0"]
    for t in range(T):                                        # teacher-forced per step, 1e-5
        _load(env, rep(prev_pos), rep(prev_vel), rep(g["ideal_shape"]), rep(g["ideal_vel"]), np.zeros(B))
        for i, a in enumerate(world.agents):
            if not g["scripted"][i]:
                a.action.u = torch.as_tensor(g["acts"][t, i])[None].expand(B, 2)     # RAW action
        world.step()
        pos, vel = world.get_state()
        for b in (0, B - 1):
            np.testing.assert_allclose(_np(pos)[b], g["pos"][t], rtol=0, atol=ATOL)
            np.testing.assert_allclose(_np(vel)[b], g["vel"][t], rtol=0, atol=ATOL)
        u = _np(world.action_u)[0][g["scripted"]]
        np.testing.assert_allclose(u, g["u_scripted"][t], rtol=0, atol=ATOL)
        obs = np.stack([_np(sc.observation(a, world))[0] for a in world.agents])
        np.testing.assert_allclose(obs, g["obs"][t], rtol=0, atol=ATOL)
        prev_pos, prev_vel = g["pos"][t], g["vel"][t]
    assert calls == [(B, 2)] * (2 * T)                        # once per scripted agent and step, on batched views
    # free-running over the fixture's horizon stays inside 1e-4 (fp32 trajectory)
    _load(env, rep(g["pos0"]), rep(g["vel0"]), rep(g["ideal_shape"]), rep(g["ideal_vel"]), np.zeros(B))
    for t in range(T):
        for i, a in enumerate(world.agents):
            if not g["scripted"][i]:
                a.action.u = torch.as_tensor(g["acts"][t, i])[None].expand(B, 2)
        world.step()
    np.testing.assert_allclose(_np(world.get_state()[0])[0], g["pos"][-1], rtol=0, atol=1e-4)


@pytest.mark.parametrize("N,B", [(5, 60), (27, 33), (70, 7), (130, 4)])
def test_random_entity_flags_against_oracle(N, B):
    """Random movable / collide / ghost flags with soft and hard walls over lane-group and whole-workgroup agent counts:
    one teacher-forced step vs the fp64 oracle on the same fp32 inputs."""
    from formation_gym.core import Wall
    rs = np.random.RandomState(77 + N)
    env = _make(N, B)
    movable = rs.uniform(size=N) > 0.25; collide = rs.uniform(size=N) > 0.25; ghost = rs.uniform(size=N) > 0.5
    movable[0] = True
    mass = rs.uniform(0.5, 3.0, N)
    for a, mv, cl, gh, m in zip(env.world.agents, movable, collide, ghost, mass):
        a.movable = bool(mv); a.collide = bool(cl); a.ghost = bool(gh); a.initial_mass = float(m)
    walls = [("V", -0.3, (-1.0, 1.0), 0.1, True), ("H", 0.1, (-0.5, 0.5), 0.1, False)]
    env.world.walls = [Wall(o, ax, ep, w, hard=h) for (o, ax, ep, w, h) in walls]
    f32 = lambda x: np.asarray(x, dtype=np.float32).astype(np.float64)
    spread = 0.4 * np.sqrt(N / 27.0)
    pos = f32(rs.uniform(-spread, spread, (B, N, 2))); vel = f32(rs.uniform(-0.5, 0.5, (B, N, 2)))
    shape = rs.uniform(-1, 1, (B, N, 2)); shape = f32(shape - shape.mean(1, keepdims=True))
    ivel = f32(rs.uniform(-1, 1, (B, 2)))
    act = rs.uniform(-1, 1, (B, N, 2)).astype(np.float32)
    _load(env, pos, vel, shape, ivel, np.zeros(B))
    env.world.action_u.copy_(torch.as_tensor(act))
    env.world.step()                                                     # the World API: immovable agents are allowed here
    out_t = dict(obs=torch.empty((B, N, 6 * N), device="cuda"), reward=torch.empty((B, N), device="cuda"),
                 indiv=torch.empty((B, N), device="cuda"))
    env.scenario.observe_batch(env.world, out_t)
    st = dict(pos=pos, vel=vel, ideal_shape=shape, ideal_vel=ivel, step=np.zeros(B, dtype=np.int32))
    new, out = O.step_hd(st, act.astype(np.float64), mass=f32(mass), movable=movable, collide=collide, ghost=ghost, walls=walls)
    p_, v_ = env.world.get_state()
    np.testing.assert_allclose(_np(p_), new["pos"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(_np(v_), new["vel"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(_np(out_t["obs"]), out["obs"], rtol=0, atol=ATOL)
    ok = out["cnt_margin"] > 1e-5
    np.testing.assert_allclose(_np(out_t["indiv"])[ok], out["indiv"][ok], rtol=0, atol=ATOL)
    assert (out["cnt"][:, ~collide] == 0).all() and out["cnt"].sum() > 0


def _scn_env(g, scenario, B, movable=True):
    """A landmark-scenario env with the fixture's per-agent mass / size / max_speed, flags and walls."""
    from formation_gym.core import Wall
    N = g["acts"].shape[2]
    env = _make(N, B, scenario)
    for i, a in enumerate(env.world.agents):
        a.initial_mass = float(g["agent_mass"][i]); a.size = float(g["agent_size"][i])
        ms = g["agent_max_speed"][i]
        a.max_speed = None if np.isnan(ms) else float(ms)
        a.collide = bool(g["agent_collide"][i]); a.ghost = bool(g["agent_ghost"][i])
        if movable and "agent_movable" in g:
            a.movable = bool(g["agent_movable"][i])
    env.world.walls = ([Wall(o, ax, ep, w) for (o, ax, ep, w) in O.GOLDEN_WALLS] +
                       [Wall(o, ax, ep, w, hard=False) for (o, ax, ep, w) in O.GOLDEN_SOFT_WALLS])
    return env


def _scn_load(env, g, t, L, rep=1):
    """State before step t of the fixture (teacher forcing), every fixture env `rep` times."""
    src = (lambda k: np.repeat(g[k + "0"], rep, 0)) if t == 0 else (lambda k: np.repeat(g[k][t - 1], rep, 0))
    env.world.set_state(src("pos"), src("vel"))
    lm, lmvel = src("lm"), src("lmvel")
    env.world.landmark_pos.copy_(torch.as_tensor(lm[:, :L], dtype=torch.float32))
    if lm.shape[1] > L:
        env.world.obstacle_pos.copy_(torch.as_tensor(lm[:, L:], dtype=torch.float32))
        env.world.obstacle_vel.copy_(torch.as_tensor(lmvel[:, L:], dtype=torch.float32))
    env.world.step_count.fill_(t)


def _scn_check(env, g, t, L, obs, indiv, rep=1):
    pos, vel = env.world.get_state()
    np.testing.assert_allclose(_np(pos)[::rep], g["pos"][t], rtol=0, atol=ATOL)
    np.testing.assert_allclose(_np(vel)[::rep], g["vel"][t], rtol=0, atol=ATOL)
    if g["lm"].shape[2] > L:
        np.testing.assert_allclose(_np(env.world.obstacle_pos)[::rep], g["lm"][t][:, L:], rtol=0, atol=ATOL)
        np.testing.assert_allclose(_np(env.world.obstacle_vel)[::rep], g["lmvel"][t][:, L:], rtol=0, atol=ATOL)
    np.testing.assert_allclose(_np(obs)[::rep], g["obs"][t], rtol=0, atol=ATOL)
    bad = np.abs(_np(indiv)[::rep] - g["indiv"][t]) > ATOL
    if bad.any():                                            # integers: excuse only a pair sitting on its threshold
        sz = g["agent_size"]
        N = len(sz)
        PD = np.sqrt(((g["pos"][t][:, :, None] - g["pos"][t][:, None]) ** 2).sum(-1)) + 10 * np.eye(N)
        assert np.abs(PD - (sz[:, None] + sz[None, :])).min() < 1e-5, "individual reward mismatch away from a threshold"


@pytest.mark.parametrize("scenario,name", [("formation_hd_obs_env", "obst_n5_flags"), ("basic_formation_env", "basic_n4_flags")])
def test_landmark_scenarios_with_flagged_agents(golden, scenario, name):
    """Agents that do not collide and ghosts (core.py:54-58, 292-293, 326-327) with per-agent mass / size among hard and soft
    walls in the landmark scenarios, through env.step: column 6 of FgParams.agent_props in the run-time-count scenario kernel
    (these were stepped like ordinary agents, then refused; now honoured).  Teacher-forced against the reference at
    1e-5; a K-step launch equals single steps bit for bit."""
    g = golden(name)
    T, B, N = g["acts"].shape[:3]
    L = int(g["num_landmarks"])
    env = _scn_env(g, scenario, B)
    free = g["indiv"][..., ~g["agent_collide"]]
    for t in range(T):
        _scn_load(env, g, t, L)
        obs, rew, done, info = env.step(torch.as_tensor(g["acts"][t]).cuda())
        _scn_check(env, g, t, L, obs, info["individual_reward"])
        np.testing.assert_array_equal(done.cpu().numpy(), g["done"][t])
    # the agent that does not collide collects no penalty: its reward is the formation term alone, the best of its env
    assert (free >= g["indiv"].max(-1, keepdims=True) - 1e-9).all() and (g["indiv"].min(-1) < g["indiv"].max(-1) - 0.5).any()
    a, b = _scn_env(g, scenario, B), _scn_env(g, scenario, B)
    for e in (a, b):
        _scn_load(e, g, 0, L)
    acts = torch.as_tensor(g["acts"][:8]).cuda().contiguous()
    o_seq, r_seq, d_seq, i_seq = b.rollout(acts)
    for t in range(8):
        o, r, d, i = a.step(acts[t])
        assert torch.equal(o, o_seq[t]) and torch.equal(r, r_seq[t]) and torch.equal(i["individual_reward"], i_seq["individual_reward"][t])


@pytest.mark.parametrize("scenario,name", [("formation_hd_obs_env", "obst_n5_immovable"), ("formation_hd_partial_env", "partial_n6_immovable")])
def test_landmark_scenarios_with_an_immovable_agent(golden, scenario, name):
    """An immovable agent (core.py:231, 266-267, 294-295, 319-321) in a landmark scenario: it keeps position and velocity,
    pushes its neighbours and the falling obstacles with the plain force (no mass ratio).  The reference's env.step asserts on
    a silent immovable agent (environment.py:236) and so does this one; the fixture was made through core.py's World API
    (world.step, then Scenario.observation / reward per agent) - its counterpart here is the scenario's batch step."""
    g = golden(name)
    T, _, N = g["acts"].shape[:3]
    L = int(g["num_landmarks"])
    rep = 3
    env = _scn_env(g, scenario, rep)
    with pytest.raises(AssertionError):
        env.step(torch.zeros((rep, N, 2), device="cuda"))
    frozen = ~g["agent_movable"]
    for t in range(T):
        _scn_load(env, g, t, L, rep)
        act = torch.as_tensor(np.repeat(g["acts"][t], rep, 0)).cuda().contiguous()
        env.scenario.step_batch(env.world, act, env._out)
        _scn_check(env, g, t, L, env._out["obs"], env._out["indiv"], rep)
        pos, vel = env.world.get_state()
        assert np.array_equal(_np(pos)[:, frozen], np.repeat(g["pos0"], rep, 0)[:, frozen].astype(np.float32))
        assert np.array_equal(_np(vel)[:, frozen], np.repeat(g["vel0"], rep, 0)[:, frozen].astype(np.float32))
    # the same steps with that agent movable end elsewhere: the flag is what the kernel honoured
    other = _scn_env(g, scenario, 1, movable=False)
    _scn_load(other, g, 0, L)
    other.scenario.step_batch(other.world, torch.as_tensor(g["acts"][0]).cuda().contiguous(), other._out)
    assert np.abs(_np(other.world.get_state()[0]) - g["pos"][0]).max() > 1e-3


@pytest.mark.parametrize("dim_c", [1, 3, 5])
def test_update_agent_state_with_other_comm_widths(dim_c):
    """World.update_agent_state (core.py:279-286) for dim_c != 2: state.c = action.c for non-silent agents (+ c_noise N(0,1)),
    zeros for silent ones - `fg_update_comm_dim`; pair 0 of the noise equals what dim_c = 2 draws."""
    B, N = 7, 6
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
    world = env.world
    world.dim_c = dim_c
    for i, a in enumerate(world.agents):
        a.silent = i % 3 == 0
        a.c_noise = None
    gen = torch.Generator(device="cuda"); gen.manual_seed(dim_c)
    want = torch.rand((B, N, dim_c), device="cuda", generator=gen)
    for i, a in enumerate(world.agents):
        if not a.silent:
            a.action.c = want[:, i]
    world.update_agent_state()
    for i, a in enumerate(world.agents):
        got = a.state.c
        assert tuple(got.shape) == (B, dim_c)
        assert torch.equal(got, torch.zeros_like(got) if a.silent else want[:, i])
    # with noise: deterministic per (seed, step), about N(0, c_noise^2) around the action, silent agents still zero
    for a in world.agents:
        a.c_noise = 0.5
    world.update_agent_state(seed=3)
    c1 = world.comm_c.clone()
    world.update_agent_state(seed=3)
    assert torch.equal(c1, world.comm_c)
    talk = [i for i, a in enumerate(world.agents) if not a.silent]
    d = (c1[:, talk] - want[:, talk]).flatten()
    assert 0.2 < float(d.std()) < 0.9 and float(d.abs().max()) < 3.0 and float(c1[:, [0, 3]].abs().max()) == 0.0
    with pytest.raises(NotImplementedError):
        env.scenario.params(world)                        # the fused kernels' communication block is dim_c = 2
