"""Device-side auto-reset of the landmark scenarios (basic_formation_env, formation_hd_partial_env,
formation_hd_partial_range_env, formation_hd_obs_env): the vec-env worker's rule (env_wrappers.py:14-18) inside the step
launch, and the standalone masked reset `fg_reset_scenario` that makes the same draws.  Run with `pytest -m gpu`."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gym-formation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
from oracle import formation_oracle as O   # noqa: E402

pytestmark = pytest.mark.gpu

SCENARIOS = [("basic_formation_env", None, 3), ("formation_hd_partial_env", "partial", 5),
             ("formation_hd_partial_range_env", "range", 4), ("formation_hd_obs_env", "obstacle", 4),
             ("formation_hd_obs_env", "obstacle", 70)]          # 70 agents + 3 obstacles: one env per workgroup


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _state(env):
    w = env.world
    return [w.pos_x, w.pos_y, w.vel_x, w.vel_y, w.landmark_pos, w.obstacle_pos, w.obstacle_vel, w.step_count]


def _pair(scenario, N, B, seed):
    import formation_gym
    envs = []
    for _ in range(2):
        env = formation_gym.make_env(scenario, False, N, num_envs=B, device="cuda:0")
        env.seed(seed)
        env.reset()
        envs.append(env)
    return envs


@pytest.mark.parametrize("scenario,kind,N", SCENARIOS)
def test_fused_auto_reset_equals_step_then_masked_reset(scenario, kind, N):
    """step with auto-reset == step without, then fg_reset_scenario on the finished envs with the same counter offset,
    then the observation of the fresh state: bit for bit, state and outputs; reward / done are the finished step's."""
    B = 37 if N < 64 else 9
    a, b = _pair(scenario, N, B, seed=5)
    a.auto_reset = True
    W = int(a.world.world_length)
    gen = torch.Generator(device="cuda"); gen.manual_seed(N)
    step0 = torch.where(torch.arange(B, device="cuda") % 3 == 0, W - 2, 3).to(torch.int32)
    for e in (a, b):
        e.world.step_count.copy_(step0)
    finished_any = False
    for t in range(3):
        act = (torch.rand((B, N, 2), generator=gen, device="cuda") * 2 - 1).contiguous()
        off = b._launch_rng_offset()
        assert off == a._launch_rng_offset()
        oa, ra, da, ia = a.step(act)
        ob, rb, db, ib = b.step(act)
        assert torch.equal(ra, rb) and torch.equal(da, db)
        assert torch.equal(ia["individual_reward"], ib["individual_reward"])
        mask = db.all(dim=1)
        assert torch.equal(mask, db.any(dim=1))
        ob = ob.clone()
        if bool(mask.any()):
            finished_any = True
            assert t == 1 and torch.equal(mask.cpu(), (torch.arange(B) % 3 == 0))
            b.scenario.reset_device(b.world, mask=mask.to(torch.uint8), rng_offset=off)
            fresh = {"obs": torch.empty_like(ob)}
            b.scenario.observe_batch(b.world, fresh)
            ob[mask] = fresh["obs"][mask]
        assert torch.equal(oa, ob), "observations differ at step %d" % t
        for x, y in zip(_state(a), _state(b)):
            assert torch.equal(x, y)
    assert finished_any
    assert int(a.world.step_count[0]) == 1 and int(a.world.step_count[1]) == 6


@pytest.mark.parametrize("scenario,kind,N", SCENARIOS[:4])
def test_device_reset_draws_and_reset_observation(scenario, kind, N):
    """What the device reset leaves behind: the distributions of reset_world (agents, landmarks U(-1,1)^2, zero
    velocities, obstacle k in [s_k, s_k+1] x [2.0, 2.5] falling), step 0, different per env and per reset; and the
    observation of the fresh state equals the oracle's on that state."""
    B = 4096
    import formation_gym
    env = formation_gym.make_env(scenario, False, N, num_envs=B, device="cuda:0")
    env.seed(3)
    env.reset()
    w, sc = env.world, env.scenario
    w.step_count.fill_(7)
    sc.reset_device(w, rng_offset=11)
    pos = torch.stack((w.pos_x, w.pos_y), -1)
    lm = w.landmark_pos
    assert (w.step_count == 0).all() and (w.vel_x == 0).all() and (w.vel_y == 0).all()
    for x in (pos, lm):
        assert (x.abs() <= 1).all()
        assert abs(float(x.mean())) < 0.02 and abs(float(x.var()) - 1.0 / 3.0) < 0.01      # U(-1,1): mean 0, var 1/3
    assert pos[0].ne(pos[1]).any() and lm[0].ne(lm[1]).any()
    # agents and landmarks are separate draws
    assert abs(float((pos[:, 0] * lm[:, 0]).mean())) < 0.02
    M = getattr(sc, "num_obstacles", 0)
    if M:
        s = np.linspace(-1.8, 1.8, M + 1)
        op = _np(w.obstacle_pos)
        for k in range(M):
            assert (op[:, k, 0] >= s[k] - 1e-6).all() and (op[:, k, 0] <= s[k + 1] + 1e-6).all()
            assert abs(op[:, k, 0].mean() - 0.5 * (s[k] + s[k + 1])) < 0.02
        assert (op[..., 1] >= 2.0).all() and (op[..., 1] <= 2.5).all() and abs(op[..., 1].mean() - 2.25) < 0.01
        assert torch.equal(w.obstacle_vel, torch.tensor(sc.OBSTACLE_VEL, device="cuda").expand(B, M, 2))
    before = pos.clone()
    mask = (torch.arange(B, device="cuda") % 2 == 0).to(torch.uint8)
    sc.reset_device(w, mask=mask, rng_offset=12)                       # another offset: other draws; masked-out envs untouched
    pos2 = torch.stack((w.pos_x, w.pos_y), -1)
    assert torch.equal(pos2[1::2], before[1::2]) and pos2[0::2].ne(before[0::2]).any(dim=-1).all()
    sc.reset_device(w, mask=mask, rng_offset=12)                       # same offset: same draws
    assert torch.equal(torch.stack((w.pos_x, w.pos_y), -1), pos2)
    out = {"obs": torch.empty((B, N, sc.obs_dim(w)), device="cuda")}
    sc.observe_batch(w, out)
    p64, v64, l64 = _np(pos2), np.zeros((B, N, 2)), _np(w.landmark_pos)
    if kind is None:
        want = O.observation_basic(p64, v64, l64)
    else:
        want = O.observation_scn(kind, p64, v64, l64, _np(w.obstacle_pos), O.ScnParams(kind))
    np.testing.assert_allclose(_np(out["obs"]), want, rtol=0, atol=1e-6)


@pytest.mark.parametrize("scenario,kind,N", SCENARIOS[:4])
def test_vec_env_device_mode_runs_episodes_of_the_landmark_scenarios(scenario, kind, N):
    """FormationVecEnv(reset_mode='device') over two episodes: done exactly at the episode ends, the observation that
    comes back with it is the reset observation, the next episode starts from a different state."""
    import formation_gym
    from formation_gym.vec_env import FormationVecEnv
    B = 64
    env = formation_gym.make_env(scenario, False, N, num_envs=B, device="cuda:0")
    env.seed(2)
    venv = FormationVecEnv(env, reset_mode="device")
    obs = venv.reset()
    W = int(env.world.world_length)
    start = torch.stack((env.world.pos_x, env.world.pos_y), -1).clone()
    act = torch.zeros((B, N, 2), device="cuda")
    starts = [start]
    for t in range(1, 2 * W + 1):
        obs, rew, done, info = venv.step(act)
        assert bool(done.all()) == (t % W == 0) and bool(done.any()) == (t % W == 0)
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
        if t % W == 0:
            assert (env.world.step_count == 0).all()
            fresh = torch.stack((env.world.pos_x, env.world.pos_y), -1).clone()
            assert fresh.ne(starts[-1]).any(dim=-1).all()
            starts.append(fresh)
            want = {"obs": torch.empty_like(obs)}
            env.scenario.observe_batch(env.world, want)
            assert torch.equal(obs, want["obs"])                        # the RESET observation
    assert len(starts) == 3 and starts[1].ne(starts[2]).any()


@pytest.mark.parametrize("scenario,kind,N", SCENARIOS)
@pytest.mark.parametrize("obs_every", [1, 3])
def test_scenario_rollout_equals_step_calls(scenario, kind, N, obs_every):
    """env.rollout on the landmark scenarios (`fg_rollout_scenario`: K steps in one launch, state on chip) == K calls of
    step, bit for bit - physics with obstacle contacts, rewards, dones, the device auto-resets in the middle of the
    launch, every obs_every-th observation; through internally allocated and through caller-owned (bound) buffers."""
    B = 45 if N < 64 else 7
    K = 9
    a, b = _pair(scenario, N, B, seed=8)
    W = int(a.world.world_length)
    step0 = ((torch.arange(B, device="cuda") * 5) % W).to(torch.int32)        # episodes end at different steps of the launch
    for e in (a, b):
        e.auto_reset = True
        e.world.step_count.copy_(step0)
        if kind == "obstacle":                                                # agents under the falling obstacles: contacts
            e.world.pos_y.add_(1.6)
    gen = torch.Generator(device="cuda"); gen.manual_seed(N + obs_every)
    out = None
    for rnd in range(3):
        acts = (torch.rand((K, B, N, 2), generator=gen, device="cuda") * 2 - 1).contiguous()
        if rnd == 1:                                                          # caller-owned buffers from here on
            D = b._out["obs"].shape[-1]
            f = dict(dtype=torch.float32, device="cuda")
            out = dict(obs=torch.empty((K // obs_every, B, N, D), **f), reward=torch.empty((K, B, N), **f),
                       indiv=torch.empty((K, B, N), **f), done=torch.zeros((K, B, N), dtype=torch.uint8, device="cuda"))
        obs, rew, done, info = b.rollout(acts, out=out, obs_every=obs_every)
        assert obs.shape[0] == K // obs_every
        for k in range(K):
            o, r, d, i = a.step(acts[k])
            assert torch.equal(r, rew[k]) and torch.equal(d, done[k]), (rnd, k)
            assert torch.equal(i["individual_reward"], info["individual_reward"][k])
            if (k + 1) % obs_every == 0:
                assert torch.equal(o, obs[k // obs_every]), "observations differ at step %d of launch %d" % (k, rnd)
        for x, y in zip(_state(a), _state(b)):
            assert torch.equal(x, y)
    assert bool(done.any()) or W > 3 * K                                      # resets happened inside the launches
    if kind == "obstacle" and N < 64:
        assert float(info["individual_reward"].min()) <= -2.0                 # a collision penalty was paid: contacts were real


@pytest.mark.parametrize("scenario,kind,N", [SCENARIOS[1], SCENARIOS[3]])
def test_captured_loop_of_a_landmark_scenario_equals_the_step_loop(scenario, kind, N):
    """FormationVecEnv.capture on a landmark scenario: replays of the captured T-step loop (policy + step with device
    auto-resets, obstacles moving) == the same loop launch by launch, and capturing leaves the env untouched
    (landmarks and obstacles included)."""
    import formation_gym
    from formation_gym.vec_env import FormationVecEnv
    B, T, R = 48, 4, 3
    dev = "cuda:0"

    def fresh():
        env = formation_gym.make_env(scenario, False, N, num_envs=B, device=dev)
        env.seed(6)
        v = FormationVecEnv(env, reset_mode="device")
        v.reset()
        env.world.world_length = 5                              # short episodes: resets inside every replay
        env.world.step_count.copy_(torch.arange(B, dtype=torch.int32, device=dev) % 5)
        return v

    D = fresh().env._out["obs"].shape[-1]
    gen = torch.Generator(device=dev); gen.manual_seed(N)
    Wp = (torch.rand((D, 2), generator=gen, device=dev) - 0.5) * 0.3
    fn = lambda obs: torch.tanh(obs @ Wp)
    ref = fresh()
    obs = ref.env._out["obs"]
    want = {k: [] for k in ("obs", "rew", "done")}
    for t in range(T * R):
        obs, rew, done, info = ref.step(fn(obs).contiguous())
        want["obs"].append(obs.clone()); want["rew"].append(rew.clone()); want["done"].append(done.clone())
    v = fresh()
    before = [x.clone() for x in _state(v.env)]
    loop = v.capture(fn, T)
    for x, y in zip(before, _state(v.env)):
        assert torch.equal(x, y)
    for r in range(R):
        o, rw, d, info = loop.replay()
        torch.cuda.synchronize()
        for t in range(T):
            k = r * T + t
            assert torch.equal(o[t], want["obs"][k]) and torch.equal(rw[t], want["rew"][k]) and torch.equal(d[t], want["done"][k]), (r, t)
    for x, y in zip(_state(ref.env), _state(v.env)):
        assert torch.equal(x, y)
    assert any(bool(x.any()) for x in want["done"])


@pytest.mark.parametrize("scenario,kind,N", [SCENARIOS[0], SCENARIOS[3]])
def test_scenario_rollout_with_world_options_equals_step_calls(scenario, kind, N):
    """The World options of core.py (max_speed :271-276, accel :236, u_noise :232-233, walls :325-362) inside a K-step
    launch of a landmark scenario: the motor noise of step k is keyed by the launch's offset + k, exactly what k step
    calls use, so rollout == steps bit for bit, with auto-resets in the middle."""
    from formation_gym.core import Wall
    B, K = 33, 7
    a, b = _pair(scenario, N, B, seed=12)
    W = int(a.world.world_length)
    for e in (a, b):
        e.auto_reset = True
        e.world.step_count.copy_(((torch.arange(B, device="cuda") * 3) % W).to(torch.int32))
        for ag in e.world.agents:
            ag.max_speed, ag.accel, ag.u_noise = 0.7, 3.5, 0.3
        e.world.walls = [Wall(o, ax, ep, w) for (o, ax, ep, w) in O.GOLDEN_WALLS]
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    for rnd in range(2):
        acts = (torch.rand((K, B, N, 2), generator=gen, device="cuda") * 2 - 1).contiguous()
        obs, rew, done, info = b.rollout(acts)
        for k in range(K):
            o, r, d, i = a.step(acts[k])
            assert torch.equal(o, obs[k]) and torch.equal(r, rew[k]) and torch.equal(d, done[k]), (rnd, k)
        for x, y in zip(_state(a), _state(b)):
            assert torch.equal(x, y)
    speed = torch.stack((a.world.vel_x, a.world.vel_y), -1).norm(dim=-1)
    assert float(speed.max()) <= 0.7 * (1 + 1e-5)
    # the noise is there: the same actions from the same state without it end elsewhere
    c, _ = _pair(scenario, N, B, seed=12)
    c.world.step_count.copy_(((torch.arange(B, device="cuda") * 3) % W).to(torch.int32))
    for ag in c.world.agents:
        ag.max_speed, ag.accel = 0.7, 3.5
    c.world.walls = [Wall(o, ax, ep, w) for (o, ax, ep, w) in O.GOLDEN_WALLS]
    c.auto_reset = True
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    acts = (torch.rand((K, B, N, 2), generator=gen, device="cuda") * 2 - 1).contiguous()
    c.rollout(acts)
    d_, _ = _pair(scenario, N, B, seed=12)
    d_.world.step_count.copy_(((torch.arange(B, device="cuda") * 3) % W).to(torch.int32))
    for ag in d_.world.agents:
        ag.max_speed, ag.accel, ag.u_noise = 0.7, 3.5, 0.3
    d_.world.walls = [Wall(o, ax, ep, w) for (o, ax, ep, w) in O.GOLDEN_WALLS]
    d_.auto_reset = True
    d_.rollout(acts)
    assert c.world.pos_x.ne(d_.world.pos_x).any()


@pytest.mark.parametrize("scenario,kind,N", [SCENARIOS[1], SCENARIOS[3]])
def test_scenario_shards_reproduce_the_global_batch_across_resets(scenario, kind, N):
    """Three shards of a landmark-scenario batch (formation_gym/sharding.py: the env-batch cut a multi-GPU run makes),
    stepped through an episode end with device auto-reset, equal the one-process batch bit for bit: the device draws
    are keyed by the GLOBAL env index (FgParams.env_index_base), so results do not depend on the number of GPUs."""
    from formation_gym import sharding
    G, K = 50, 6
    whole, lo, hi = sharding.make_env_shard(scenario, N, G, seed=4, rank=0, world_size=1, local_rank=0)
    assert (lo, hi) == (0, G)
    parts = [sharding.make_env_shard(scenario, N, G, seed=4, rank=r, world_size=3, local_rank=0) for r in range(3)]
    W = int(whole.world.world_length)
    step0 = ((torch.arange(G, device="cuda") * 7) % W).to(torch.int32)
    envs = [(whole, 0, G)] + parts
    for e, l, h in envs:
        e.reset()
        e.auto_reset = True
        e.world.step_count.copy_(step0[l:h])
    for e, l, h in parts:
        for s, t in zip(_state(whole), _state(e)):
            assert torch.equal(s[l:h], t)                      # same host reset streams
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    acts = (torch.rand((2 * K, G, N, 2), generator=gen, device="cuda") * 2 - 1).contiguous()
    saw_done = False
    for k in range(K):                                         # step launches
        o, r, d, i = whole.step(acts[k])
        saw_done = saw_done or bool(d.any())
        for e, l, h in parts:
            o2, r2, d2, i2 = e.step(acts[k, l:h].contiguous())
            assert torch.equal(o2, o[l:h]) and torch.equal(r2, r[l:h]) and torch.equal(d2, d[l:h]), (k, l)
    obs, rew, done, info = whole.rollout(acts[K:])             # and one rollout launch
    saw_done = saw_done or bool(done.any())
    for e, l, h in parts:
        obs2, rew2, done2, _ = e.rollout(acts[K:, l:h].contiguous())
        assert torch.equal(obs2, obs[:, l:h]) and torch.equal(rew2, rew[:, l:h]) and torch.equal(done2, done[:, l:h])
        for s, t in zip(_state(whole), _state(e)):
            assert torch.equal(s[l:h], t)
    assert saw_done


@pytest.mark.parametrize("scenario,kind,N", SCENARIOS)
def test_device_mt19937_reset_of_the_landmark_scenarios_is_bit_exact(scenario, kind, N):
    """fg_reset_scenario_mt continues each env's legacy NumPy stream on the GPU: three consecutive resets (all envs, a masked
    half, the envs whose episode is over) equal the host path `reset_world` - the reference's draws - exactly, obstacles'
    np.random.uniform([s_k, 2.0], [s_k+1, 2.5]) included."""
    B = 37
    dev_env, host_env = _pair(scenario, N, B, seed=91)
    dev_env.scenario.upload_mt_streams(dev_env.world)          # both envs have drawn their first reset: the streams are aligned
    for rep in range(3):
        if rep == 0:
            host_env.scenario.reset_world(host_env.world)
            dev_env.scenario.reset_mt(dev_env.world)
        elif rep == 1:
            m = np.zeros(B, dtype=bool); m[::2] = True
            host_env.scenario.reset_world(host_env.world, env_mask=m)
            dev_env.scenario.reset_mt(dev_env.world, torch.as_tensor(m.astype(np.uint8)).cuda())
        else:
            L = int(host_env.world.world_length)
            steps = torch.as_tensor(np.where(np.arange(B) % 3 == 1, L, 2).astype(np.int32)).cuda()
            for e in (dev_env, host_env):
                e.world.step_count.copy_(steps)
            host_env.scenario.reset_world(host_env.world, env_mask=(np.arange(B) % 3 == 1))
            dev_env.scenario.reset_mt_done(dev_env.world)
        for x, y in zip(_state(dev_env), _state(host_env)):
            assert torch.equal(x, y), (scenario, rep)


@pytest.mark.parametrize("scenario,kind,N", SCENARIOS[:4])
def test_vec_env_device_mt_of_the_landmark_scenarios_matches_host_mode(scenario, kind, N):
    """FormationVecEnv(reset_mode='device_mt') for the landmark scenarios: multi-episode rollouts equal the 'host' reset mode
    bit for bit (observations incl. the reset observation, pre-reset rewards and dones, the streams' continuation)."""
    import formation_gym
    from formation_gym.vec_env import FormationVecEnv
    B, T, L = 11, 9, 4
    phase = (np.arange(B) * 3) % L
    envs = []
    for mode in ("device_mt", "host"):
        e = formation_gym.make_env(scenario, False, N, num_envs=B, device="cuda:0")
        e.seed(33)
        v = FormationVecEnv(e, reset_mode=mode)
        v.reset()
        e.world.world_length = L
        e.world.step_count.copy_(torch.as_tensor(phase.astype(np.int32)))
        v.ts[:] = phase
        envs.append((e, v))
    rs = np.random.RandomState(5)
    for t in range(T):
        act = torch.as_tensor(rs.uniform(-1, 1, (B, N, 2)).astype(np.float32)).cuda()
        outs = [v.step(act) for (_, v) in envs]
        for k in range(3):
            assert torch.equal(torch.as_tensor(outs[0][k]), torch.as_tensor(outs[1][k])), (scenario, t, k)
        for x, y in zip(_state(envs[0][0]), _state(envs[1][0])):
            assert torch.equal(x, y)
