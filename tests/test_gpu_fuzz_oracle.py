"""Randomised differential test of `fg_step_hd` against the fp64 oracle (the CPU restatement of the reference, pinned by
the reference's own outputs in tests/golden/): seeded random draws of agent count (3 ... 300, specialised and run-time
instantiations, fused and split single steps), batch size, crowding (from sparse to heavy contact), World options (walls,
max_speed, accel), and episode phase; two steps teacher-forced from the same fp32 state.  Every fp32 bound is 1e-5 abs
(north_star); integer-valued terms (collision counts, done) are exact except where a pair sits within 1e-5 of a threshold."""
import os

import numpy as np
import pytest
import torch

from oracle import formation_oracle as O

pytestmark = pytest.mark.gpu
SEEDS = range(int(os.environ.get("FG_FUZZ_SEEDS", "100")))     # FG_FUZZ_SEEDS=300: 20 s on one MI355X
ATOL = 1e-5


def _np(t):
    return t.detach().double().cpu().numpy()


@pytest.mark.parametrize("seed", SEEDS)
def test_random_step_equals_oracle(seed):
    import formation_gym
    from formation_gym.core import Wall
    rs = np.random.RandomState(5000 + seed)
    N = int(rs.choice([3, 4, 5, 8, 9, 16, 27, 27, 33, 64, 65, 81, 100, 243, 300]))
    B = int(rs.choice([1, 3, 17, 64, 130])) if N <= 100 else int(rs.choice([1, 2, 5]))
    crowd = float(rs.choice([1.0, 0.6, 0.3, 0.12]))
    opts = {}
    if rs.rand() < 0.4:
        if rs.rand() < 0.6:
            opts["max_speed"] = float(rs.uniform(0.2, 0.8))
        if rs.rand() < 0.5:
            opts["accel"] = float(rs.uniform(1.5, 4.0))
        if rs.rand() < 0.5:
            opts["walls"] = True
    f32 = lambda x: np.asarray(x, dtype=np.float32).astype(np.float64)
    st = O.reset_hd(rs.randint(0, 100000, B), N)
    state = dict(pos=f32(st["pos"] * crowd), vel=f32(rs.uniform(-0.4, 0.4, (B, N, 2))), ideal_shape=f32(st["ideal_shape"]),
                 ideal_vel=f32(st["ideal_vel"]), step=rs.randint(0, 99, B).astype(np.int32))
    state["step"][rs.rand(B) < 0.25] = 99                                 # done flips in the first step
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
    for ag in env.world.agents:
        ag.max_speed = opts.get("max_speed")
        ag.accel = opts.get("accel")
    if opts.get("walls"):
        env.world.walls = [Wall(o, ax, ep, w) for (o, ax, ep, w) in O.GOLDEN_WALLS]
    assign = bool(rs.rand() < 0.3)                                          # the index-emitting instantiations
    if assign:
        env.enable_assignments(True)
    P = O.HdParams()
    tag = "seed %d: N=%d B=%d crowd=%.2f opts=%s assign=%s" % (seed, N, B, crowd, opts, assign)
    for t in range(2):
        env.world.set_state(state["pos"], state["vel"])
        env.scenario.set_formation(env.world, state["ideal_shape"], state["ideal_vel"])
        env.world.step_count.copy_(torch.as_tensor(state["step"], dtype=torch.int32))
        act = f32(rs.uniform(-1, 1, (B, N, 2)))
        obs, rew, done, info = env.step(torch.as_tensor(act, dtype=torch.float32).cuda())
        new, out = O.step_hd(state, act, P, max_speed=opts.get("max_speed"), accel=opts.get("accel"),
                             walls=O.GOLDEN_WALLS if opts.get("walls") else None)
        pos, vel = env.world.get_state()
        # a pair at distance ~0 (coincident agents) gives NaN in both; the draws above never produce one
        np.testing.assert_allclose(_np(pos), new["pos"], rtol=0, atol=ATOL, err_msg=tag)
        np.testing.assert_allclose(_np(vel), new["vel"], rtol=0, atol=ATOL, err_msg=tag)
        np.testing.assert_allclose(_np(obs), out["obs"], rtol=0, atol=ATOL, err_msg=tag)
        np.testing.assert_array_equal(done.cpu().numpy(), out["done"], err_msg=tag)
        ok = out["cnt_margin"] > 1e-5                                       # envs without a pair on the collision threshold
        # |indiv| grows with the collision count (up to ~N in the crowded draws): fp32 resolution there is above 1e-5
        scale = np.maximum(1.0, np.abs(out["indiv"][ok]) / 16.0)
        assert (np.abs(_np(info["individual_reward"])[ok] - out["indiv"][ok]) <= ATOL * scale).all(), tag
        np.testing.assert_allclose(_np(rew)[ok][..., 0], out["reward"][ok][..., 0], rtol=2e-6, atol=ATOL, err_msg=tag)
        if assign:                                                           # indices: on the GPU's own fp32 state, near-ties excused
            r = O.reward_hd(_np(pos), _np(vel), state["ideal_shape"], state["ideal_vel"], P)
            for key, gap in (("near_lm", "gap_lm"), ("near_ag", "gap_ag")):
                got = env._out[key].cpu().numpy()
                bad = got != r[key]
                assert (r[gap][bad] < 1e-6).all(), tag + " " + key
            tie = r["hd_gap"].min(1) < 1e-6
            np.testing.assert_array_equal(env._out["hd_idx"].cpu().numpy()[~tie], r["hd_idx"][~tie], err_msg=tag)
        state = dict(new, pos=f32(new["pos"]), vel=f32(new["vel"]))
        state["step"] = np.where(out["done"][:, 0], 0, new["step"]).astype(np.int32)


@pytest.mark.parametrize("seed", SEEDS)
def test_random_landmark_scenario_step_equals_oracle(seed):
    """The same for fg_step_scenario / fg_step_basic: scenario kind, agent count 2 ... 90 (lane-group and whole-workgroup
    envs), batch, crowding; two steps teacher-forced from the same fp32 state."""
    import formation_gym
    rs = np.random.RandomState(9000 + seed)
    scenario, kind = [("formation_hd_partial_env", "partial"), ("formation_hd_partial_range_env", "range"),
                      ("formation_hd_obs_env", "obstacle"), ("basic_formation_env", "basic")][rs.randint(4)]
    N = int(rs.choice([3, 4, 5, 7, 9, 13, 16, 30, 61, 62, 64, 65, 90]))
    if kind == "partial":
        N = max(N, 5)                                                      # num_obs = 3 ring neighbours + itself
    B = int(rs.choice([1, 3, 16, 17, 70]))
    crowd = float(rs.choice([1.0, 0.5, 0.2]))
    env = formation_gym.make_env(scenario, False, N, num_envs=B, device="cuda:0")
    P = O.BasicParams() if kind == "basic" else O.ScnParams(kind)
    L, M = P.num_landmarks, getattr(P, "num_obstacles", 0)
    f32 = lambda x: np.asarray(x, dtype=np.float32).astype(np.float64)
    state = dict(pos=f32(rs.uniform(-1, 1, (B, N, 2)) * crowd), vel=f32(rs.uniform(-0.3, 0.3, (B, N, 2))),
                 landmarks=f32(rs.uniform(-1, 1, (B, L, 2))), step=rs.randint(0, P.world_length - 1, B).astype(np.int32))
    if kind != "basic":
        state["obst_pos"] = f32(rs.uniform(-0.8, 0.8, (B, M, 2)))
        state["obst_vel"] = f32(np.tile(np.array(P.obstacle_vel), (B, M, 1)))
    tag = "seed %d: %s N=%d B=%d crowd=%.1f" % (seed, kind, N, B, crowd)
    for t in range(2):
        env.world.set_state(state["pos"], state["vel"])
        env.world.landmark_pos.copy_(torch.as_tensor(state["landmarks"], dtype=torch.float32))
        if M:
            env.world.obstacle_pos.copy_(torch.as_tensor(state["obst_pos"], dtype=torch.float32))
            env.world.obstacle_vel.copy_(torch.as_tensor(state["obst_vel"], dtype=torch.float32))
        env.world.step_count.copy_(torch.as_tensor(state["step"], dtype=torch.int32))
        act = f32(rs.uniform(-1, 1, (B, N, 2)))
        obs, rew, done, info = env.step(torch.as_tensor(act, dtype=torch.float32).cuda())
        new, out = (O.step_basic(state, act, P) if kind == "basic" else O.step_scn(kind, state, act, P))
        pos, vel = env.world.get_state()
        np.testing.assert_allclose(_np(pos), new["pos"], rtol=0, atol=ATOL, err_msg=tag)
        np.testing.assert_allclose(_np(vel), new["vel"], rtol=0, atol=ATOL, err_msg=tag)
        np.testing.assert_allclose(_np(obs), out["obs"], rtol=0, atol=ATOL, err_msg=tag)
        np.testing.assert_array_equal(done.cpu().numpy(), out["done"], err_msg=tag)
        if M:
            np.testing.assert_allclose(_np(env.world.obstacle_pos), new["obst_pos"], rtol=0, atol=ATOL, err_msg=tag)
            np.testing.assert_allclose(_np(env.world.obstacle_vel), new["obst_vel"], rtol=0, atol=ATOL, err_msg=tag)
        PD = np.sqrt(((new["pos"][:, :, None] - new["pos"][:, None]) ** 2).sum(-1)) + (0 if kind == "basic" else 10 * np.eye(N))
        ok = np.abs(PD - P.collide_thresh).min((1, 2)) > 1e-5
        if M:
            OD = np.sqrt(((new["pos"][:, :, None] - new["obst_pos"][:, None]) ** 2).sum(-1))
            ok &= np.abs(OD - (P.agent_size + P.obstacle_size)).min((1, 2)) > 1e-5
        scale = np.maximum(1.0, np.abs(out["indiv"][ok]) / 16.0)
        assert (np.abs(_np(info["individual_reward"])[ok] - out["indiv"][ok]) <= ATOL * scale).all(), tag
        np.testing.assert_allclose(_np(rew)[ok][..., 0], np.repeat(out["shared"][:, None], N, 1)[ok], rtol=2e-6, atol=ATOL, err_msg=tag)
        state = dict(new, pos=f32(new["pos"]), vel=f32(new["vel"]))
        if M:
            state["obst_pos"] = f32(new["obst_pos"]); state["obst_vel"] = f32(new["obst_vel"])
        state["step"] = np.minimum(new["step"], P.world_length - 1).astype(np.int32)
