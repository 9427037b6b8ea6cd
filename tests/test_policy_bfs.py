"""Batched ezpolicy / get_action_BFS (product, torch) against the actions the
reference produced (tests/golden/policy_n*.npz).  CPU float64, no GPU."""
import numpy as np
import pytest
import torch

from oracle import formation_oracle as O


@pytest.mark.parametrize("name", ["policy_n3", "policy_n9", "policy_n27"])
def test_bfs_matches_reference(golden, name):
    from formation_gym.policy_bfs import ezpolicy, get_action_BFS
    g = golden(name)
    N = g["pos0"].shape[0]
    for t in range(g["act"].shape[0]):
        pos = g["pos"][t - 1] if t else g["pos0"]
        vel = g["vel"][t - 1] if t else g["vel0"]
        obs = O.observation_hd(pos[None], vel[None], g["ideal_shape"][None], g["ideal_vel"][None])[0]
        act = get_action_BFS(ezpolicy, list(obs), 3)
        assert isinstance(act, list) and len(act) == N and act[0].shape == (2,)
        np.testing.assert_allclose(np.array(act), g["act"][t], rtol=0, atol=1e-9)
        # tensor calling convention, two envs at once
        ot = torch.as_tensor(np.stack([obs, obs]))
        at = get_action_BFS(ezpolicy, ot, 3)
        assert at.shape == (2, N, 2)
        np.testing.assert_allclose(at[1].numpy(), g["act"][t], rtol=0, atol=1e-9)


def test_ezpolicy_single_vector(golden):
    from formation_gym.policy_bfs import ezpolicy
    g = golden("policy_n3")
    for o, want in zip(g["obs0"], g["ez_act0"]):
        got = ezpolicy(o)
        assert got.shape == (2,)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)


def test_bfs_rejects_bad_agent_count():
    from formation_gym.policy_bfs import ezpolicy, get_action_BFS
    with pytest.raises(AssertionError):
        get_action_BFS(ezpolicy, [np.zeros(24)] * 4, 3)
