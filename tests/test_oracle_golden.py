"""Pin the CPU oracle (oracle/formation_oracle.py) to the golden fixtures that
tests/golden/make_golden.py captured from the real reference, plus scipy's
published directed_hausdorff example.  CPU only."""
import os

import numpy as np
import pytest

from oracle import formation_oracle as O

HD_CASES = ["hd_n3", "hd_n9", "hd_n27", "hd_n81", "hd_n9_crowd", "hd_n27_crowd",
            "hd_n81_crowd", "hd_n243", "hd_n4", "hd_n10", "hd_n5", "hd_n6_crowd", "hd_n16_crowd", "hd_n50", "hd_n100_crowd", "hd_n3_done"]
TOL = 1e-12


def _state0(g):
    B = g["pos0"].shape[0]
    return dict(pos=g["pos0"], vel=g["vel0"], ideal_shape=g["ideal_shape"],
                ideal_vel=g["ideal_vel"], step=np.zeros(B, dtype=np.int32))


@pytest.mark.parametrize("name", HD_CASES)
def test_hd_free_running_matches_reference(golden, name):
    g = golden(name)
    st = _state0(g)
    T = g["acts"].shape[0]
    np.testing.assert_allclose(
        O.observation_hd(st["pos"], st["vel"], st["ideal_shape"], st["ideal_vel"]),
        g["obs0"], rtol=0, atol=TOL)
    for t in range(T):
        st, out = O.step_hd(st, g["acts"][t].astype(np.float64))
        # fp64 free-running: contacts amplify 1e-16 rounding, so allow growth
        tol = 1e-9 if "crowd" in name or name == "hd_n243" else 1e-11
        np.testing.assert_allclose(st["pos"], g["pos"][t], rtol=0, atol=tol)
        np.testing.assert_allclose(st["vel"], g["vel"][t], rtol=0, atol=tol * 10)
        np.testing.assert_array_equal(out["done"], g["done"][t])


@pytest.mark.parametrize("name", HD_CASES)
def test_hd_teacher_forced_step(golden, name):
    """Each step re-seeded from the reference's own previous state."""
    g = golden(name)
    T = g["acts"].shape[0]
    B = g["pos0"].shape[0]
    prev_pos, prev_vel = g["pos0"], g["vel0"]
    for t in range(T):
        st = dict(pos=prev_pos, vel=prev_vel, ideal_shape=g["ideal_shape"],
                  ideal_vel=g["ideal_vel"], step=np.full(B, t, dtype=np.int32))
        st, out = O.step_hd(st, g["acts"][t].astype(np.float64))
        np.testing.assert_allclose(st["pos"], g["pos"][t], rtol=0, atol=TOL)
        np.testing.assert_allclose(st["vel"], g["vel"][t], rtol=0, atol=1e-11)
        # rewards/indices from the reference's own post-step state
        r = O.reward_hd(g["pos"][t], g["vel"][t], g["ideal_shape"], g["ideal_vel"], O.HdParams())
        np.testing.assert_allclose(r["indiv"], g["indiv"][t], rtol=0, atol=TOL)
        np.testing.assert_allclose(r["shared"][:, None].repeat(g["shared"].shape[2], 1),
                                   g["shared"][t], rtol=1e-13, atol=TOL)
        np.testing.assert_allclose(r["hd"], g["hd"][t], rtol=0, atol=TOL)
        np.testing.assert_array_equal(r["hd_idx"], g["hd_idx"][t])
        np.testing.assert_array_equal(r["near_lm"], g["near_lm"][t])
        np.testing.assert_array_equal(r["near_ag"], g["near_ag"][t])
        np.testing.assert_array_equal(r["cnt"], g["cnt"][t])
        np.testing.assert_array_equal(out["done"], g["done"][t])
        prev_pos, prev_vel = g["pos"][t], g["vel"][t]


@pytest.mark.parametrize("name", HD_CASES)
def test_hd_observation_layout(golden, name):
    g = golden(name)
    for t in g["obs_steps"]:
        obs = O.observation_hd(g["pos"][t - 1], g["vel"][t - 1], g["ideal_shape"], g["ideal_vel"])
        np.testing.assert_allclose(obs, g["obs_t%d" % t], rtol=0, atol=TOL)
        N = obs.shape[1]
        assert obs.shape[2] == 6 * N
        assert np.all(obs[:, :, 2 * N:4 * N - 2] == 0.0)      # silent agents: comm zeros


def test_hd_done_flips_at_world_length(golden):
    g = golden("hd_n3_done")
    first = np.argmax(g["done"].reshape(g["done"].shape[0], -1).any(1)) + 1
    assert first == O.HdParams.world_length == 100
    assert g["done"][99:].all() and not g["done"][:99].any()


def test_basic_env_matches_reference(golden):
    g = golden("basic_n3")
    P = O.BasicParams()
    assert int(g["world_length"]) == P.world_length
    assert int(g["num_landmarks"]) == P.num_landmarks
    assert float(g["agent_size"]) == P.agent_size
    st = dict(pos=g["pos0"], vel=g["vel0"], landmarks=g["landmarks"], step=np.zeros(1, dtype=np.int32))
    np.testing.assert_allclose(O.observation_basic(st["pos"], st["vel"], st["landmarks"]),
                               g["obs0"], rtol=0, atol=TOL)
    assert g["obs0"].shape[-1] == int(g["obs_dim"]) == 18
    for t in range(g["acts"].shape[0]):
        st, out = O.step_basic(st, g["acts"][t].astype(np.float64), P)
        np.testing.assert_allclose(st["pos"], g["pos"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out["obs"], g["obs"][t], rtol=0, atol=1e-9)
        np.testing.assert_allclose(out["indiv"], g["indiv"][t], rtol=0, atol=1e-9)
        np.testing.assert_allclose(out["reward"][..., 0], g["shared"][t], rtol=0, atol=1e-9)
        np.testing.assert_array_equal(out["done"], g["done"][t])
    assert g["done"][49:].all() and not g["done"][:49].any()
    # every agent carries the constant -1 of its self-"collision" (basic_formation_env.py:49-51)
    assert (out["cnt"] >= 1).all()


def test_reset_stream(golden):
    g = golden("reset")
    for seed, N in g["cases"]:
        key = "s%d_n%d" % (seed, N)
        st = O.reset_hd([seed], N)
        np.testing.assert_array_equal(st["pos"][0], g[key + "_pos"])
        np.testing.assert_array_equal(st["ideal_vel"][0], g[key + "_ivel"])
        np.testing.assert_allclose(st["ideal_shape"][0], g[key + "_shape"], rtol=0, atol=1e-15)
        assert (g[key + "_vel"] == 0).all()
        obs = O.observation_hd(st["pos"], st["vel"], st["ideal_shape"], st["ideal_vel"])
        np.testing.assert_allclose(obs[0], g[key + "_obs"], rtol=0, atol=1e-15)
        assert int(g[key + "_obs_dim"]) == 6 * N
        assert int(g[key + "_share_obs_dim"]) == 6 * N * N
        assert int(g[key + "_world_length"]) == 100
        # observation's side effect re-centres the landmarks on the agents (:40-44)
        np.testing.assert_allclose(g[key + "_landmarks"].mean(0), st["pos"][0].mean(0), atol=1e-12)


def test_generate_shape(golden):
    g = golden("shapes")
    for L in range(4):
        np.testing.assert_allclose(O.generate_shape(L), g["layer%d" % L], rtol=0, atol=1e-15)
        assert O.generate_shape(L).shape == (3 ** (L + 1), 2)
    with pytest.raises(AssertionError):
        O.generate_shape(4)


def test_hausdorff_known_answer(golden):
    g = golden("hausdorff_kat")
    d, i, j = O.directed_hausdorff_bruteforce(g["u"], g["v"])
    assert d == float(g["d_uv"]) == 2.23606797749979           # scipy docstring value
    d2, _, _ = O.directed_hausdorff_bruteforce(g["v"], g["u"])
    assert d2 == float(g["d_vu"]) == 3.0


@pytest.mark.parametrize("name,per", [("policy_n3", 3), ("policy_n9", 3), ("policy_n27", 3), ("policy_n81", 3),
                                      ("policy_n8_per2", 2), ("policy_n16_per4", 4), ("policy_n5_per5", 5)])
def test_bfs_policy_matches_reference(golden, name, per):
    g = golden(name)
    N = g["pos0"].shape[0]
    P = O.HdParams()
    pos, vel = g["pos0"][None], g["vel0"][None]
    for t in range(g["act"].shape[0]):
        prev_pos = g["pos"][t - 1][None] if t else pos
        prev_vel = g["vel"][t - 1][None] if t else vel
        obs = O.observation_hd(prev_pos, prev_vel, g["ideal_shape"][None], g["ideal_vel"][None])[0]
        act = np.array(O.get_action_bfs(O.ezpolicy, list(obs), per))
        np.testing.assert_allclose(act, g["act"][t], rtol=0, atol=1e-9)
    if N == 3:
        ez = np.array([O.ezpolicy(o) for o in g["obs0"]])
        np.testing.assert_allclose(ez, g["ez_act0"], rtol=0, atol=1e-12)
    # decision margins (what the fp32 GPU test may excuse): one per agent, non-negative, and never larger than
    # the margin of the top-level problem of the sub-group the agent belongs to
    m = O.bfs_margins(list(obs), per)
    assert m.shape == (N,) and (m >= 0).all() and np.isfinite(m).all()
    tops = np.array([O.ezpolicy_margin(x) for x in _top_level_inputs(obs, per)])
    assert (m <= np.repeat(tops, N // per) + 1e-15).all()


def test_bfs_shape_test_keeps_the_reference_quirk():
    """__init__.py:55-56 compares a float log ratio with an integer: 243 = 3^5 agents are rejected by the
    reference itself [probed: np.log(243)/np.log(3) = 4.999999999999999]."""
    obs = [np.zeros(6 * 243)] * 243
    with pytest.raises(AssertionError):
        O.get_action_bfs(O.ezpolicy, obs, 3)
    with pytest.raises(AssertionError):
        O.get_action_bfs(O.ezpolicy, [np.zeros(60)] * 10, 3, strict=False)


def _top_level_inputs(obs, per):
    """The `per` observation vectors the top level of get_action_BFS hands to the policy (__init__.py:62-77)."""
    seen = []

    def spy(inp):
        seen.append(np.array(inp))
        return O.ezpolicy(inp)

    O.get_action_bfs(spy, list(obs), per)
    return seen[:per]


def test_benchmark_data_matches_reference(golden):
    """Scenario.benchmark_data (formation_hd_env.py:97-117) as make_env(benchmark=True) evaluates it."""
    g = golden("benchmark_n9")
    assert g["info_keys"].all()                       # the flag changes nothing in step()'s return (environment.py:130-133)
    P = O.HdParams()
    for t in range(g["pos"].shape[0]):
        r = O.reward_hd(g["pos"][t][None], g["vel"][t][None], g["ideal_shape"][None], g["ideal_vel"][None], P)
        np.testing.assert_allclose(r["indiv"][0], g["indiv"][t], rtol=0, atol=1e-12)
        bd = O.benchmark_data_hd(g["pos"][t][None], g["lm"][t][None], r["indiv"], P)
        np.testing.assert_allclose(bd["reward"][0], g["b_reward"][t], rtol=0, atol=1e-12)
        np.testing.assert_array_equal(bd["collisions"][0], g["b_collisions"][t])
        np.testing.assert_allclose(bd["min_dists"][0], g["b_min_dists"][t], rtol=0, atol=1e-12)
        np.testing.assert_array_equal(bd["occupied_landmarks"][0], g["b_occupied"][t])
        # the landmarks the reference measures against were re-centred on the agents by observation() (:40-44)
        np.testing.assert_allclose(g["lm"][t], g["ideal_shape"] + g["pos"][t].mean(0), rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", ["hd_n3", "hd_n9_crowd", "hd_n27_crowd"])
def test_port_env_matches_reference(golden, name):
    """The per-env faithful port (the timed cpu_baseline) against the fixtures."""
    g = golden(name)
    N = g["pos0"].shape[1]
    for b in range(min(2, g["pos0"].shape[0])):
        env = O.PortEnv(N)
        env.load(g["pos0"][b], g["vel0"][b], g["ideal_shape"][b], g["ideal_vel"][b])
        for t in range(min(10, g["acts"].shape[0])):
            obs_n, rew_n, done_n, info_n = env.step(list(g["acts"][t, b].astype(np.float64)))
            pos = np.array([a.pos for a in env.agents])
            np.testing.assert_allclose(pos, g["pos"][t, b], rtol=0, atol=1e-10)
            np.testing.assert_allclose([i["individual_reward"] for i in info_n], g["indiv"][t, b],
                                       rtol=0, atol=1e-9)
            np.testing.assert_allclose(rew_n[0][0], g["shared"][t, b, 0], rtol=1e-12, atol=1e-9)
            if (t + 1) in g["obs_steps"]:
                np.testing.assert_allclose(np.array(obs_n), g["obs_t%d" % (t + 1)][b], rtol=0, atol=1e-10)


def test_port_env_reset_stream(golden):
    g = golden("reset")
    env = O.PortEnv(9)
    env.seed(7)
    obs = env.reset()
    np.testing.assert_allclose(np.array(obs), g["s7_n9_obs"], rtol=0, atol=1e-15)


SCN_CASES = [("partial", "partial_n5"), ("partial", "partial_n9_crowd"), ("partial", "partial_n3"),
             ("range", "range_n4"), ("range", "range_n7_crowd"), ("obstacle", "obst_n4"), ("obstacle", "obst_n8")]


def _scn_state(g, P, t=None):
    L = P.num_landmarks
    src = (lambda k: g[k + "0"]) if t is None else (lambda k: g[k][t])
    B = g["pos0"].shape[0]
    return dict(pos=src("pos"), vel=src("vel"), landmarks=src("lm")[:, :L], obst_pos=src("lm")[:, L:],
                obst_vel=src("lmvel")[:, L:], step=np.full(B, 0 if t is None else t + 1, dtype=np.int32))


@pytest.mark.parametrize("kind,name", SCN_CASES)
def test_remaining_scenarios_match_reference(golden, kind, name):
    g = golden(name)
    P = O.ScnParams(kind)
    assert int(g["world_length"]) == P.world_length and int(g["num_landmarks"]) == P.num_landmarks
    assert float(g["agent_size"]) == P.agent_size
    assert int(g["num_entities"]) == P.num_landmarks + P.num_obstacles
    st = _scn_state(g, P)
    np.testing.assert_allclose(O.observation_scn(kind, st["pos"], st["vel"], st["landmarks"], st["obst_pos"], P),
                               g["obs0"], rtol=0, atol=1e-12)
    assert g["obs0"].shape[-1] == int(g["obs_dim"])
    T = g["acts"].shape[0]
    for t in range(T):
        prev = _scn_state(g, P, t - 1) if t else st
        new, out = O.step_scn(kind, prev, g["acts"][t].astype(np.float64), P)
        np.testing.assert_allclose(new["pos"], g["pos"][t], rtol=0, atol=1e-11)
        np.testing.assert_allclose(new["vel"], g["vel"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(new["obst_pos"], g["lm"][t][:, P.num_landmarks:], rtol=0, atol=1e-11)
        np.testing.assert_allclose(new["obst_vel"], g["lmvel"][t][:, P.num_landmarks:], rtol=0, atol=1e-11)
        np.testing.assert_allclose(out["obs"], g["obs"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out["indiv"], g["indiv"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out["reward"][..., 0], g["shared"][t], rtol=1e-12, atol=1e-10)
        np.testing.assert_array_equal(out["done"], g["done"][t])


@pytest.mark.parametrize("kind,name", [("partial", "partial_n5"), ("range", "range_n4"), ("obstacle", "obst_n4")])
def test_remaining_scenarios_reset_stream(golden, kind, name):
    g = golden(name)
    N = g["pos0"].shape[1]
    P = O.ScnParams(kind)
    for b in range(g["pos0"].shape[0]):
        st = O.reset_scn(kind, int(g["seed"]) + 1000 * b, N)
        np.testing.assert_array_equal(st["pos"][0], g["pos0"][b])
        np.testing.assert_array_equal(st["landmarks"][0], g["lm0"][b][:P.num_landmarks])
        np.testing.assert_array_equal(st["obst_pos"][0], g["lm0"][b][P.num_landmarks:])
        np.testing.assert_array_equal(st["obst_vel"][0], g["lmvel0"][b][P.num_landmarks:])


@pytest.mark.parametrize("name,opts", [("hd_n9_options", dict(max_speed=0.6, accel=3.0, walls=O.GOLDEN_WALLS)),
                                        ("hd_n27_walls", dict(walls=O.GOLDEN_WALLS))])
def test_world_options_match_reference(golden, name, opts):
    """max_speed / accel / walls: World features no reference scenario switches on (f4)."""
    g = golden(name)
    B = g["pos0"].shape[0]
    prev_pos, prev_vel = g["pos0"], g["vel0"]
    for t in range(g["acts"].shape[0]):
        st = dict(pos=prev_pos, vel=prev_vel, ideal_shape=g["ideal_shape"], ideal_vel=g["ideal_vel"],
                  step=np.full(B, t, dtype=np.int32))
        st, out = O.step_hd(st, g["acts"][t].astype(np.float64), **opts)
        np.testing.assert_allclose(st["pos"], g["pos"][t], rtol=0, atol=1e-11)
        np.testing.assert_allclose(st["vel"], g["vel"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out["indiv"], g["indiv"][t], rtol=0, atol=1e-10)
        prev_pos, prev_vel = g["pos"][t], g["vel"][t]


def _params_from_fixture(g):
    """HdParams carrying the non-default World constants a `*_constants` fixture was made with."""
    P = O.HdParams()
    P.dt = float(g["world_dt"]); P.damping = float(g["world_damping"])
    P.contact_force = float(g["world_contact_force"]); P.contact_margin = float(g["world_contact_margin"])
    P.mass = float(g["world_mass"]); P.agent_size = float(g["world_size"]); P.world_length = int(g["world_world_length"])
    return P


@pytest.mark.parametrize("name", ["hd_n9_constants", "hd_n27_constants"])
def test_non_default_world_constants_match_reference(golden, name):
    """dt, damping, contact force / margin, agent mass and size, episode length away from the
    defaults of core.py:119-139 (set on the reference's World before the rollout)."""
    g = golden(name)
    P = _params_from_fixture(g)
    B = g["pos0"].shape[0]
    prev_pos, prev_vel = g["pos0"], g["vel0"]
    assert g["cnt"].sum() > 0 and g["done"].any() and not g["done"].all()
    for t in range(g["acts"].shape[0]):
        st = dict(pos=prev_pos, vel=prev_vel, ideal_shape=g["ideal_shape"], ideal_vel=g["ideal_vel"],
                  step=np.full(B, t, dtype=np.int32))
        st, out = O.step_hd(st, g["acts"][t].astype(np.float64), P)
        np.testing.assert_allclose(st["pos"], g["pos"][t], rtol=0, atol=1e-11)
        np.testing.assert_allclose(st["vel"], g["vel"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out["indiv"], g["indiv"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out["reward"][..., 0], g["shared"][t], rtol=1e-12, atol=1e-10)
        np.testing.assert_array_equal(out["done"], g["done"][t])
        if (t + 1) in g["obs_steps"]:
            np.testing.assert_allclose(out["obs"], g["obs_t%d" % (t + 1)], rtol=0, atol=1e-10)
        prev_pos, prev_vel = g["pos"][t], g["vel"][t]


def hetero_opts(g, walls=False):
    """World options of a `*_masses` fixture: per-agent mass / size / accel / max_speed (NaN = None)."""
    o = dict(mass=g["agent_mass"], size=g["agent_size"], accel=g["agent_accel"], max_speed=g["agent_max_speed"])
    if walls:
        o["walls"] = O.GOLDEN_WALLS
    return o


@pytest.mark.parametrize("name,walls", [("hd_n9_masses", False), ("hd_n27_masses", True)])
def test_per_agent_mass_size_and_options_match_reference(golden, name, walls):
    """Agents of different mass, size, accel and max_speed (set on the reference's agents before the rollout):
    force_ratio m_b / m_a (core.py:314-317), per-pair contact and penalty distances (core.py:307,
    formation_hd_env.py:119-121), per-agent gains (core.py:236, environment.py:219-220) and speed clamps (:271-276)."""
    g = golden(name)
    B = g["pos0"].shape[0]
    opts = hetero_opts(g, walls)
    assert g["cnt"].sum() > 0 and np.ptp(g["agent_mass"]) > 1 and np.isnan(g["agent_accel"]).any() and (~np.isnan(g["agent_accel"])).any()
    prev_pos, prev_vel = g["pos0"], g["vel0"]
    for t in range(g["acts"].shape[0]):
        st = dict(pos=prev_pos, vel=prev_vel, ideal_shape=g["ideal_shape"], ideal_vel=g["ideal_vel"],
                  step=np.full(B, t, dtype=np.int32))
        st, out = O.step_hd(st, g["acts"][t].astype(np.float64), **opts)
        np.testing.assert_allclose(st["pos"], g["pos"][t], rtol=0, atol=1e-11)
        np.testing.assert_allclose(st["vel"], g["vel"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out["indiv"], g["indiv"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out["reward"][..., 0], g["shared"][t], rtol=1e-12, atol=1e-10)
        np.testing.assert_array_equal(out["cnt"], g["cnt"][t])
        if (t + 1) in g["obs_steps"]:
            np.testing.assert_allclose(out["obs"], g["obs_t%d" % (t + 1)], rtol=0, atol=1e-10)
        prev_pos, prev_vel = g["pos"][t], g["vel"][t]


def test_non_silent_agents_match_reference(golden):
    """World.update_agent_state (core.py:279-286) and the communication block of the observation
    (formation_hd_env.py:48-51), driven through core.py's own API; the reference's env.step cannot take
    non-silent agents at all (IndexError at environment.py:231), which the fixture records."""
    g = golden("hd_n5_comm")
    assert str(g["env_step_raises"]).startswith("IndexError")
    pos, vel = g["pos0"][None], g["vel0"][None]
    for t in range(g["acts"].shape[0]):
        st = dict(pos=pos, vel=vel, ideal_shape=g["ideal_shape"][None], ideal_vel=g["ideal_vel"][None],
                  step=np.zeros(1, dtype=np.int32))
        c = O.update_comm(g["comm"][t][None], g["silent"])
        np.testing.assert_array_equal(c[0], g["c"][t])
        st, out = O.step_hd(st, g["acts"][t][None].astype(np.float64), comm=c)
        np.testing.assert_allclose(st["pos"][0], g["pos"][t], rtol=0, atol=1e-11)
        np.testing.assert_allclose(out["obs"][0], g["obs"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out["indiv"][0], g["indiv"][t], rtol=0, atol=1e-10)
        pos, vel = st["pos"], st["vel"]
    assert (g["obs"][:, 0, 2 * 5:4 * 5 - 2] != 0).any()


@pytest.mark.skipif(not os.path.isdir(os.environ.get("FG_REFERENCE", "/root/reference")),
                    reason="the reference is only present in the build container")
def test_committed_fixtures_equal_the_generator_output(tmp_path):
    """Fixture / script drift guard: re-run tests/golden/make_golden.py against the real reference into a temp
    directory and demand key-for-key, bit-for-bit equality with the committed .npz files."""
    import glob
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, os.path.join(root, "tests", "golden", "make_golden.py")], check=True, timeout=1500,
                   env=dict(os.environ, FG_GOLDEN_OUT=str(tmp_path), PYTHONDONTWRITEBYTECODE="1"), capture_output=True)
    committed = sorted(os.path.basename(f) for f in glob.glob(os.path.join(root, "tests", "golden", "*.npz")))
    made = sorted(os.path.basename(f) for f in glob.glob(str(tmp_path / "*.npz")))
    assert committed == made
    for name in committed:
        with np.load(os.path.join(root, "tests", "golden", name)) as a, np.load(str(tmp_path / name)) as b:
            assert sorted(a.files) == sorted(b.files), name
            for k in a.files:
                if a[k].dtype.kind in "US":
                    assert str(a[k]) == str(b[k]), (name, k)
                else:
                    np.testing.assert_array_equal(a[k], b[k], err_msg="%s[%s]" % (name, k))


ACT_MODES = [("act_onehot5_n3", O.ACT_ONEHOT5), ("act_index_n9", O.ACT_INDEX), ("act_argmax_n3", O.ACT_ARGMAX)]


@pytest.mark.parametrize("name,mode", ACT_MODES)
def test_action_modes_match_reference(golden, name, mode):
    """The non-default branches of _set_action (environment.py:187-216), free-running from the
    reference's reset state; also what the reference leaves in the caller's action arrays."""
    g = golden(name)
    st = dict(pos=g["pos0"][None], vel=g["vel0"][None], ideal_shape=g["ideal_shape"][None],
              ideal_vel=g["ideal_vel"][None], step=np.zeros(1, dtype=np.int32))
    for t in range(g["acts"].shape[0]):
        u = O.decode_actions(g["acts"][t], mode)
        st, out = O.step_hd(st, u[None])
        np.testing.assert_allclose(st["pos"][0], g["pos"][t], rtol=0, atol=1e-11)
        np.testing.assert_allclose(st["vel"][0], g["vel"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out["indiv"][0], g["indiv"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out["reward"][0, :, 0], g["shared"][t], rtol=1e-12, atol=1e-10)
        np.testing.assert_allclose(out["obs"][0], g["obs"][t], rtol=0, atol=1e-10)
        if mode == O.ACT_ARGMAX:          # overwritten with the one-hot, then scaled through the u view
            np.testing.assert_array_equal(g["acts_after"][t], 5.0 * u)
        else:
            np.testing.assert_array_equal(g["acts_after"][t], g["acts"][t].astype(np.float64))
    assert int(g["action_space_n"]) == (5 if mode == O.ACT_ONEHOT5 else -1)


def test_entity_flags_match_reference(golden):
    """Entity.movable / collide / ghost (core.py:54-58) and a soft wall: the oracle free-runs on the reference's trajectory."""
    g = golden("hd_n6_immovable")
    assert str(g["env_step_raises"]).startswith("AssertionError")          # environment.py:236 for a silent immovable agent
    P = O.HdParams()
    pos, vel = g["pos0"][None], g["vel0"][None]
    for t in range(g["acts"].shape[0]):
        pos, vel = O.physics_step(pos, vel, g["acts"][t][None].astype(np.float64), P, mass=g["mass"], movable=g["movable"], collide=g["collide"])
        np.testing.assert_allclose(pos[0], g["pos"][t], rtol=0, atol=1e-12)
        np.testing.assert_allclose(vel[0], g["vel"][t], rtol=0, atol=1e-12)
        r = O.reward_hd(pos, vel, g["ideal_shape"][None], g["ideal_vel"][None], P, collide=g["collide"])
        np.testing.assert_allclose(r["indiv"][0], g["indiv"][t], rtol=0, atol=1e-12)
        np.testing.assert_allclose(O.observation_hd(pos, vel, g["ideal_shape"][None], g["ideal_vel"][None])[0], g["obs"][t], rtol=0, atol=1e-12)
    g = golden("hd_n9_flags")
    walls = O.GOLDEN_WALLS + [w + (False,) for w in O.GOLDEN_SOFT_WALLS]
    T, B = g["acts"].shape[:2]
    st = dict(pos=g["pos0"], vel=g["vel0"], ideal_shape=g["ideal_shape"], ideal_vel=g["ideal_vel"], step=np.zeros(B, dtype=np.int32))
    opts = dict(mass=g["agent_mass"], size=g["agent_size"], accel=g["agent_accel"], max_speed=g["agent_max_speed"], walls=walls,
                collide=g["agent_collide"], ghost=g["agent_ghost"])
    P = O.HdParams(); P.agent_size = float(g["agent_size"][0])
    for t in range(T):
        st, out = O.step_hd(st, g["acts"][t].astype(np.float64), P, **opts)
        np.testing.assert_allclose(st["pos"], g["pos"][t], rtol=0, atol=1e-11)
        np.testing.assert_allclose(out["indiv"], g["indiv"][t], rtol=0, atol=1e-10)
        if (t + 1) in g["obs_steps"]:
            np.testing.assert_allclose(out["obs"], g["obs_t%d" % (t + 1)], rtol=0, atol=1e-10)
    # the soft wall and the flags matter in this fixture: an oracle blind to them leaves the reference's trajectory
    st2 = dict(pos=g["pos0"], vel=g["vel0"], ideal_shape=g["ideal_shape"], ideal_vel=g["ideal_vel"], step=np.zeros(B, dtype=np.int32))
    blind = dict(opts); blind.pop("collide"); blind.pop("ghost")
    for t in range(T):
        st2, _ = O.step_hd(st2, g["acts"][t].astype(np.float64), P, **blind)
    assert np.abs(st2["pos"] - g["pos"][-1]).max() > 1e-3


def scripted_u(p_own, v_own, p_lead):
    """tests/golden/make_golden.py scripted_u: the deterministic scripted agent of fixture hd_n6_scripted"""
    return 0.6 * np.stack([-p_own[..., 1], p_own[..., 0]], -1) - 0.3 * v_own + 0.2 * (p_lead - p_own)


def test_scripted_agents_match_reference(golden):
    """Agent.action_callback (core.py:210-211): the callback's action.u is used as it is, the policy agents' raw actions
    are scaled by the sensitivity (environment.py:216-221).  The oracle free-runs on the reference's trajectory."""
    g = golden("hd_n6_scripted")
    P = O.HdParams()
    scripted = g["scripted"]
    assert scripted.sum() == 2
    pos, vel = g["pos0"][None], g["vel0"][None]
    for t in range(g["acts"].shape[0]):
        act = g["acts"][t][None].astype(np.float64)
        u = scripted_u(pos[0][scripted], vel[0][scripted], pos[0][0][None])
        np.testing.assert_allclose(u, g["u_scripted"][t], rtol=0, atol=1e-12)
        act[0][scripted] = u
        pos, vel = O.physics_step(pos, vel, act, P, mass=g["mass"], scripted=scripted)
        np.testing.assert_allclose(pos[0], g["pos"][t], rtol=0, atol=1e-12)
        np.testing.assert_allclose(vel[0], g["vel"][t], rtol=0, atol=1e-12)
        np.testing.assert_allclose(O.observation_hd(pos, vel, g["ideal_shape"][None], g["ideal_vel"][None])[0], g["obs"][t], rtol=0, atol=1e-12)
    # the flag matters: with the sensitivity applied to the scripted agents too the oracle leaves the reference's trajectory
    pos2, vel2 = g["pos0"][None], g["vel0"][None]
    for t in range(g["acts"].shape[0]):
        act = g["acts"][t][None].astype(np.float64)
        act[0][scripted] = scripted_u(pos2[0][scripted], vel2[0][scripted], pos2[0][0][None])
        pos2, vel2 = O.physics_step(pos2, vel2, act, P, mass=g["mass"])
    assert np.abs(pos2[0] - g["pos"][-1]).max() > 1e-3


ALL_WALLS = O.GOLDEN_WALLS + [w + (False,) for w in O.GOLDEN_SOFT_WALLS]


@pytest.mark.parametrize("kind,name", [("obstacle", "obst_n5_flags"), ("obstacle", "obst_n5_immovable"), ("partial", "partial_n6_immovable")])
def test_landmark_scenarios_with_flagged_agents_match_reference(golden, kind, name):
    """Agents that do not collide, ghosts, an immovable agent (core.py:54-58) among walls in the landmark scenarios: the
    restatement against the reference (obst_n5_flags through env.step, the immovable fixtures through core.py's World API)."""
    g = golden(name)
    P = O.ScnParams(kind)
    T = g["acts"].shape[0]
    movable = g["agent_movable"] if "agent_movable" in g else None
    seen = 0
    for t in range(T):
        prev = _scn_state(g, P, t - 1) if t else _scn_state(g, P)
        new, out = O.step_scn(kind, prev, g["acts"][t].astype(np.float64), P, mass=g["agent_mass"], size=g["agent_size"],
                              max_speed=g["agent_max_speed"], movable=movable, collide=g["agent_collide"],
                              ghost=g["agent_ghost"], walls=ALL_WALLS)
        np.testing.assert_allclose(new["pos"], g["pos"][t], rtol=0, atol=1e-11)
        np.testing.assert_allclose(new["vel"], g["vel"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(new["obst_pos"], g["lm"][t][:, P.num_landmarks:], rtol=0, atol=1e-11)
        np.testing.assert_allclose(out["obs"], g["obs"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out["indiv"], g["indiv"][t], rtol=0, atol=1e-10)
        # the flags matter: the same step with ordinary agents goes elsewhere
        plain, _ = O.step_scn(kind, prev, g["acts"][t].astype(np.float64), P, mass=g["agent_mass"], size=g["agent_size"],
                              max_speed=g["agent_max_speed"], walls=ALL_WALLS)
        seen += int(np.abs(plain["pos"] - new["pos"]).max() > 1e-4)
    assert seen > T // 4
    if movable is not None:
        frozen = ~movable
        assert np.array_equal(g["pos"][-1][:, frozen], g["pos0"][:, frozen]) and np.array_equal(g["vel"][-1][:, frozen], g["vel0"][:, frozen])
        assert np.abs(g["vel0"][:, frozen]).min() > 0.1       # ... whatever velocity it had


def test_basic_scenario_with_flagged_agents_matches_reference(golden):
    """basic_formation_env with per-agent mass / size, an agent that does not collide (no contact force, no penalties - the self
    "collision" of :48-51 included), a ghost and the walls."""
    g = golden("basic_n4_flags")
    P = O.BasicParams()
    T, B = g["acts"].shape[:2]
    opts = dict(mass=g["agent_mass"], size=g["agent_size"], collide=g["agent_collide"], ghost=g["agent_ghost"], walls=ALL_WALLS)
    for t in range(T):
        src = (lambda k: g[k + "0"]) if t == 0 else (lambda k: g[k][t - 1])
        st = dict(pos=src("pos"), vel=src("vel"), landmarks=src("lm"), step=np.full(B, t))
        new, out = O.step_basic(st, g["acts"][t].astype(np.float64), P, **opts)
        np.testing.assert_allclose(new["pos"], g["pos"][t], rtol=0, atol=1e-11)
        np.testing.assert_allclose(new["vel"], g["vel"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out["obs"], g["obs"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out["indiv"], g["indiv"][t], rtol=0, atol=1e-10)
    cnt = np.round(-g["indiv"] - (-g["indiv"]).min(2, keepdims=True))
    assert (g["indiv"][..., ~g["agent_collide"]] > g["indiv"][..., g["agent_collide"]].max(-1, keepdims=True) - 1e-9).all()
    assert cnt.max() >= 1


@pytest.mark.parametrize("kind,name", [("obstacle", "obst_n5_masses"), ("partial", "partial_n6_masses")])
def test_landmark_scenarios_with_per_agent_tables_match_reference(golden, kind, name):
    g = golden(name)
    P = O.ScnParams(kind)
    T = g["acts"].shape[0]
    st = _scn_state(g, P)
    for t in range(T):
        prev = _scn_state(g, P, t - 1) if t else st
        new, out = O.step_scn(kind, prev, g["acts"][t].astype(np.float64), P, mass=g["agent_mass"], size=g["agent_size"],
                              max_speed=g["agent_max_speed"])
        np.testing.assert_allclose(new["pos"], g["pos"][t], rtol=0, atol=1e-11)
        np.testing.assert_allclose(new["vel"], g["vel"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out["obs"], g["obs"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out["indiv"], g["indiv"][t], rtol=0, atol=1e-10)
    assert (g["indiv"] < -1.5).any()
