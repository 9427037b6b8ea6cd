"""GPU parity of every launch path `fg_rollout_hd` can take, at the sizes they are used:

* exactly the launches bench.py times - (27 x 4096, K=20), (9 x 4096, K=20), (81 x 2048, K=20), (243 x 8192, K=4):
  full grids, full LDS residency, the double-buffer wrap over 20 steps, device auto-reset at mixed episode
  phases - against K `env.step` calls bit for bit, plus the fp64 oracle teacher-forced from the GPU's own state
  on a 32-env sample at the first and the last step of the launch;
* World options (walls, max_speed, accel): a rollout launch must integrate the same physics as `env.step`
  (they exist only in step_kernel's options instantiation, which then runs the K-loop);
* agent counts without a specialised kernel (run-time N) through the K-loop, with auto-reset.
"""
import numpy as np
import pytest
import torch

from oracle import formation_oracle as O

pytestmark = pytest.mark.gpu

ATOL = 1e-5


def _make(N, B):
    import formation_gym
    return formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")


def _np(t):
    return t.detach().double().cpu().numpy()


def _pair(N, B, seed, crowd, step0):
    """Two envs in the same (device-drawn, crowded) state with mixed episode phases."""
    envs = []
    for _ in range(2):
        e = _make(N, B)
        e.scenario.seed(seed)
        e.scenario.reset_device(e.world, rng_offset=12345)
        e.world.pos_x.mul_(crowd); e.world.pos_y.mul_(crowd)
        e.world.step_count.copy_(torch.as_tensor(step0, dtype=torch.int32))
        e.auto_reset = True
        envs.append(e)
    a, b = envs
    for x, y in zip(a.world.get_state() + (a.scenario.ideal_shape,), b.world.get_state() + (b.scenario.ideal_shape,)):
        assert torch.equal(x, y)
    return a, b


def _oracle_step(state, act, sample):
    """fp64 oracle on the sampled envs of a GPU state snapshot."""
    sub = dict(pos=_np(state["pos"][sample]), vel=_np(state["vel"][sample]),
               ideal_shape=_np(state["shape"][sample]), ideal_vel=_np(state["ivel"][sample]),
               step=state["step"][sample].cpu().numpy().astype(np.int32))
    return O.step_hd(sub, _np(act[sample]))


def _snapshot(env):
    pos, vel = env.world.get_state()
    return dict(pos=pos.clone(), vel=vel.clone(), shape=env.scenario.ideal_shape.clone(),
                ivel=env.scenario.ideal_vel.clone(), step=env.world.step_count.clone())


@pytest.mark.parametrize("N,B,K", [(27, 4096, 20), (9, 4096, 20), (81, 2048, 20), (243, 8192, 4)])
def test_bench_launches_equal_single_steps_and_oracle(N, B, K):
    rs = np.random.RandomState(N)
    # episode phases: a third of the envs ends its episode inside the launch (at different steps), the rest does not
    step0 = np.where(np.arange(B) % 3 == 0, 100 - 1 - (np.arange(B) // 3) % K, rs.randint(0, 100 - K, B))
    a, b = _pair(N, B, seed=3, crowd=0.45, step0=step0)
    gen = torch.Generator(device="cuda"); gen.manual_seed(N)
    acts = (torch.rand((K, B, N, 2), generator=gen, device="cuda") * 2 - 1).contiguous()
    obs, rew, done, info = b.rollout(acts)
    sample = torch.as_tensor(rs.choice(B, 32, replace=False)).cuda()
    n_done = 0
    for k in range(K):
        before = _snapshot(a) if k in (0, K - 1) else None
        o, r, d, i = a.step(acts[k])
        assert torch.equal(o, obs[k]), "observations differ at step %d" % k
        assert torch.equal(r, rew[k]) and torch.equal(d, done[k])
        assert torch.equal(i["individual_reward"], info["individual_reward"][k])
        n_done += int(d[:, 0].sum())
        if before is not None:
            new, out = _oracle_step(before, acts[k], sample)
            np.testing.assert_array_equal(d[sample].cpu().numpy(), out["done"])          # bit-exact
            live = ~out["done"][:, 0]                                                     # envs the launch did not re-draw
            pos, vel = a.world.get_state()
            np.testing.assert_allclose(_np(pos[sample])[live], new["pos"][live], rtol=0, atol=ATOL)
            np.testing.assert_allclose(_np(o[sample])[live], out["obs"][live], rtol=0, atol=2 * ATOL)   # differences of two positions
            ok = out["cnt_margin"] > 1e-5
            np.testing.assert_allclose(_np(i["individual_reward"][sample])[ok], out["indiv"][ok], rtol=0, atol=ATOL)
            np.testing.assert_allclose(_np(r[sample])[ok, :, 0], out["reward"][ok][..., 0], rtol=1e-5, atol=ATOL)
    assert B // 3 - 2 <= n_done <= B // 3 + 2                                               # those episodes really ended inside
    for x, y in zip(a.world.get_state(), b.world.get_state()):
        assert torch.equal(x, y)
    assert torch.equal(a.world.step_count, b.world.step_count)
    assert torch.equal(a.scenario.ideal_shape, b.scenario.ideal_shape) and torch.equal(a.scenario.ideal_vel, b.scenario.ideal_vel)
    assert torch.isfinite(obs).all()


@pytest.mark.parametrize("N,B,K,opts", [(9, 50, 6, dict(max_speed=0.6, accel=3.0, walls=True)),
                                         (27, 37, 5, dict(walls=True)),
                                         (27, 21, 4, dict(max_speed=0.4)),
                                         (81, 5, 3, dict(accel=2.0, walls=True)),
                                         (243, 2, 3, dict(max_speed=0.5))])
def test_rollout_with_world_options_equals_single_steps(N, B, K, opts):
    """env.rollout on a World with walls / max_speed / accel == K env.step calls, bit for bit, and the options are
    really in force (the same rollout without them gives a different trajectory)."""
    from formation_gym.core import Wall
    rs = np.random.RandomState(31 + N)
    st = O.reset_hd(rs.randint(0, 10000, B), N)
    st["pos"] *= 0.9
    acts = torch.as_tensor(rs.uniform(-1, 1, (K, B, N, 2)).astype(np.float32)).cuda()
    step0 = np.where(np.arange(B) % 4 == 0, 98, 7)
    envs = []
    for with_opts in (True, True, False):
        e = _make(N, B)
        e.world.set_state(st["pos"], st["vel"])
        e.scenario.set_formation(e.world, st["ideal_shape"], st["ideal_vel"])
        e.world.step_count.copy_(torch.as_tensor(step0, dtype=torch.int32))
        e.scenario.seed(13); e.auto_reset = True
        if with_opts:
            for ag in e.world.agents:
                ag.max_speed = opts.get("max_speed")
                ag.accel = opts.get("accel")
            if opts.get("walls"):
                e.world.walls = [Wall(o, ax, ep, w) for (o, ax, ep, w) in O.GOLDEN_WALLS]
        envs.append(e)
    a, b, plain = envs
    obs, rew, done, info = b.rollout(acts)
    for k in range(K):
        o, r, d, i = a.step(acts[k])
        assert torch.equal(o, obs[k]) and torch.equal(r, rew[k]) and torch.equal(d, done[k])
        assert torch.equal(i["individual_reward"], info["individual_reward"][k])
    for x, y in zip(a.world.get_state(), b.world.get_state()):
        assert torch.equal(x, y)
    obs_plain = plain.rollout(acts)[0]
    assert not torch.equal(obs_plain[K - 1], obs[K - 1])
    # first step against the fp64 oracle with the same options
    new, out = O.step_hd(dict(pos=np.asarray(st["pos"], np.float32).astype(np.float64), vel=st["vel"],
                              ideal_shape=np.asarray(st["ideal_shape"], np.float32).astype(np.float64),
                              ideal_vel=np.asarray(st["ideal_vel"], np.float32).astype(np.float64),
                              step=step0.astype(np.int32)), _np(acts[0]),
                         max_speed=opts.get("max_speed"), accel=opts.get("accel"),
                         walls=O.GOLDEN_WALLS if opts.get("walls") else None)
    live = ~out["done"][:, 0]
    np.testing.assert_allclose(_np(obs[0])[live], out["obs"][live], rtol=0, atol=2 * ATOL)


@pytest.mark.parametrize("N,B,K", [(10, 41, 7), (33, 9, 5), (100, 3, 4), (300, 2, 3)])
def test_rollout_with_runtime_agent_count_equals_single_steps(N, B, K):
    """Agent counts without a specialised kernel: the K-loop of the run-time-N step kernel, device auto-reset on."""
    rs = np.random.RandomState(N)
    step0 = np.where(np.arange(B) % 2 == 0, 100 - 2, 11)
    a, b = _pair(N, B, seed=8, crowd=0.4, step0=step0)
    acts = torch.as_tensor(rs.uniform(-1, 1, (K, B, N, 2)).astype(np.float32)).cuda()
    obs, rew, done, info = b.rollout(acts)
    for k in range(K):
        o, r, d, i = a.step(acts[k])
        assert torch.equal(o, obs[k]) and torch.equal(r, rew[k]) and torch.equal(d, done[k])
    assert done.any() and not done.all()
    for x, y in zip(a.world.get_state(), b.world.get_state()):
        assert torch.equal(x, y)
    assert torch.equal(a.scenario.ideal_shape, b.scenario.ideal_shape)
