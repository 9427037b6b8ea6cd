"""Host-side logic that needs no GPU: spaces, scenario plugin loading, the
reference-style action staging rules."""
import os

import numpy as np
import pytest

import formation_gym
from formation_gym import spaces


def test_box_space_mirrors_gym():
    b = spaces.Box(low=-1.0, high=+1.0, shape=(2,), dtype=np.float32)
    b.seed(0)
    s = b.sample()
    assert s.shape == (2,) and s.dtype == np.float32 and b.contains(s)
    assert not b.contains(np.array([2.0, 0.0], dtype=np.float32))
    o = spaces.Box(low=-np.inf, high=+np.inf, shape=(54,), dtype=np.float32)
    assert o.shape == (54,) and np.isinf(o.low).all()
    assert spaces.Discrete(5).contains(4) and not spaces.Discrete(5).contains(5)
    assert len(spaces.Tuple([b, o]).sample()) == 2


def test_scenario_plugins_load_by_name_and_path():
    import os
    sc = formation_gym.load_scenario("formation_hd_env")
    assert hasattr(sc, "make_world") and hasattr(sc, "reset_world")
    assert hasattr(sc, "observation") and hasattr(sc, "reward") and hasattr(sc, "benchmark_data")
    path = os.path.join(os.path.dirname(formation_gym.__file__), "envs", "basic_formation_env.py")
    assert type(formation_gym.load_scenario(path)).__name__ == "Scenario"
    # generate_shape needs no device: same nesting as the reference (3, ..., 3, 2)
    g = sc.generate_shape(2)
    assert g.shape == (3, 3, 3, 2)
    with pytest.raises(AssertionError):
        sc.generate_shape(4)


def test_generate_shape_matches_reference(golden):
    sc = formation_gym.load_scenario("formation_hd_env")
    g = golden("shapes")
    for L in range(4):
        np.testing.assert_allclose(np.asarray(sc.generate_shape(L), dtype=np.float64).reshape(-1, 2),
                                   g["layer%d" % L], rtol=0, atol=1e-15)


def test_reference_style_scenario_files_take_the_callback_adapter():
    """load_scenario: a file with only the reference's per-agent callbacks (scenario.py:4-12) is wrapped, the batched
    scenarios of envs/ are not."""
    import os
    import formation_gym
    from formation_gym.callback_scenario import CallbackScenario
    plugin = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plugins", "ring_patrol_env.py")
    sc = formation_gym.load_scenario(plugin)
    assert isinstance(sc, CallbackScenario) and "host callbacks" in sc.PATH
    user = sc._factory()
    w = user.make_world(4)                         # the user's file runs on host-mode entities (no device needed)
    assert len(w.agents) == 4 and w.agents[0].state.p_pos.shape == (2,) and w.agents[3].max_speed == 0.5
    assert len(user.observation(w.agents[0], w)) == 2 + 2 + 2 + 2 * 3 + 3
    assert not isinstance(formation_gym.load_scenario("formation_hd_env"), CallbackScenario)


def test_arena_geometry_for_the_bench_shapes():
    """placement.arena_geometry: 8-16 chunks per buffer (32 MiB - 1 GiB each); the arena is SMALL - 6 x the buffer up to
    48 GiB, at least 1.5 x the buffer (profiles/r04_place/) - never more than `mem_fraction` of the free
    memory, None when there is nothing to choose from (host arithmetic, no GPU)."""
    from formation_gym import placement
    free = 308 * 10 ** 9
    for nbytes in (1433 * 10 ** 6, 6450 * 10 ** 6, 46438 * 10 ** 6, 324 * 10 ** 6, 11627 * 10 ** 6):
        total, chunk = placement.arena_geometry(nbytes, free)
        W = -(-nbytes // chunk)
        assert chunk & (chunk - 1) == 0 and (32 << 20) <= chunk <= (1 << 30)
        assert W <= 16 or chunk == (1 << 30)
        assert 1.5 * nbytes - 1 <= total <= max(1.5 * nbytes, 48 << 30) + 1 and total <= 0.5 * free + 1 and total // chunk <= 2048
        assert total >= nbytes + 2 * chunk
    assert placement.arena_geometry(1433 * 10 ** 6, free)[0] == 6 * 1433 * 10 ** 6          # the headline buffer: 8.6 GB
    assert placement.arena_geometry(1433 * 10 ** 6, free, max_arena_bytes=192 << 30)[0] == int(0.5 * free)   # the round-3 probe, on request
    total, chunk = placement.arena_geometry(93 * 10 ** 9, 300 * 10 ** 9)        # 243 x 65536: 1.5 x the buffer
    assert chunk == 1 << 30 and 93 * 10 ** 9 + 2 * chunk <= total <= 0.9 * 300 * 10 ** 9
    assert placement.arena_geometry(93 * 10 ** 9, 96 * 10 ** 9) is None         # no room to shuffle in
    assert placement.arena_geometry(10 ** 9, 10 ** 9) is None
    # two ranks sharing a device halve the fraction
    assert placement.arena_geometry(1433 * 10 ** 6, 10 ** 10, mem_fraction=0.25)[0] <= 0.25 * 10 ** 10 + 1
    # an arena that gained nothing is followed by one four times as large, within mem_fraction of the free memory, or none
    assert placement.next_arena_bytes(10 << 30, 6 << 30, 280 << 30) == 40 << 30
    assert placement.next_arena_bytes(40 << 30, 6 << 30, 280 << 30) == 140 << 30
    assert placement.next_arena_bytes(140 << 30, 6 << 30, 280 << 30) is None
    assert placement.next_arena_bytes(87 << 30, 58 << 30, 280 << 30) is None        # 243 x 8192: already most of what is free
    assert placement.next_arena_bytes(8 << 30, 1 << 30, 20 << 30) is None


def test_tensor_plugin_callbacks_equal_the_per_agent_file():
    """tests/plugins/ring_patrol_tensor_env.py (the batched-callback contract, formation_gym/tensor_scenario.py) states the
    observation / reward of tests/plugins/ring_patrol_env.py (the reference's per-agent plugin API) on tensors: the same numbers
    on a hand-made state, on the CPU (the GPU suite holds both against the reference's fixture: test_gpu_tensor_scenario.py)."""
    import importlib.util
    import torch

    def load(name):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plugins", name + ".py")
        spec = importlib.util.spec_from_file_location("plugin_" + name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m.Scenario()

    class Obj(object):
        pass
    B, N = 4, 6
    batched, per_agent = load("ring_patrol_tensor_env"), load("ring_patrol_env")
    w = Obj(); w.num_envs = B; w.device = torch.device("cpu"); w.agents = []; w.landmarks = []
    batched.build_world(w, N)
    rs = np.random.RandomState(3)
    pos, vel = rs.uniform(-.25, .25, (B, N, 2)), rs.uniform(-1, 1, (B, N, 2))
    beacon, radius = rs.uniform(-.3, .3, (B, 1, 2)), rs.uniform(.3, .6, B)
    pos[1, 2] = pos[1, 4] + [0.01, 0.0]                                    # a pair in contact
    w.get_state = lambda: (torch.as_tensor(pos, dtype=torch.float32), torch.as_tensor(vel, dtype=torch.float32))
    w.landmark_pos = torch.as_tensor(beacon, dtype=torch.float32)
    batched.radius = torch.as_tensor(radius, dtype=torch.float32)
    obs, rew = batched.observation_batch(w).numpy(), batched.reward_batch(w).numpy()
    assert obs.shape == (B, N, 6 + 3 * (N - 1)) and rew.shape == (B, N)
    for b in range(B):
        hw = Obj(); hw.agents = []; hw.landmarks = [Obj()]
        hw.landmarks[0].state = Obj(); hw.landmarks[0].state.p_pos = beacon[b, 0]
        for i in range(N):
            a = Obj(); a.state = Obj(); a.state.p_pos = pos[b, i]; a.state.p_vel = vel[b, i]; a.size = w.agents[i].size
            hw.agents.append(a)
        per_agent.radius = radius[b]
        for i, a in enumerate(hw.agents):
            np.testing.assert_allclose(obs[b, i], per_agent.observation(a, hw), rtol=0, atol=2e-6)
            np.testing.assert_allclose(rew[b, i], per_agent.reward(a, hw), rtol=0, atol=2e-6)
