"""fp64 "parity mode" of the fused step kernel (second option): the SAME kernel source as the product
library compiled with real = double (csrc/formation_hip_f64.hip) FREE-RUNS from every fixture's initial state over
ALL of its recorded steps and must stay on the reference's float64 trajectory - positions, velocities, individual and
shared rewards, observations, done masks, landmark-index assignments.  This is what the fp32 product cannot show
beyond ~10 steps (stiff contacts amplify fp32 rounding chaotically), and it proves that the kernel's ALGORITHM - pair
loop with its far-pair cutoff, softplus contact force, Euler step, Hausdorff / collision reward pass, flat observation
writer, in-wave and cross-wave reductions - tracks core.py:206-225 / :289-322 and formation_hd_env.py:38-75.

Bound: 1e-9 abs on everything, crowded fixtures (positions scaled by 0.15-0.3, dozens of simultaneous stiff contacts
per step) included.  Measured maxima (profiles/r02_parity_errors.md): 2e-16 ... 7e-12 on positions after 25 steps
(hd_n81: summation order over 81 partners amplified by 25 steps of contacts), 5e-14 on individual rewards."""
import numpy as np
import pytest

from tests import parity_errors as PE

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", PE.HD_CASES)
def test_f64_kernel_free_runs_on_the_reference_trajectory(golden, name):
    g = golden(name)
    r = PE.free_running_f64(g)
    tol = 1e-9
    for k, v in r["err"].items():
        assert v <= tol, "%s: %s error %.3g over %d free-running steps (bound %.0e)" % (name, k, v, r["steps"], tol)
    assert r["idx_bad"]["done"] == 0
    assert r["idx_bad"]["near_lm"] == 0 and r["idx_bad"]["near_ag"] == 0      # bit-exact away from exact ties
    assert np.isfinite(r["per_step_pos"]).all()


@pytest.mark.parametrize("N,B", [(5, 9), (17, 5), (33, 3), (64, 2), (65, 2), (130, 2), (300, 1), (600, 1)])
def test_f64_kernel_against_the_oracle_at_other_agent_counts(N, B):
    """Every lane-group width and workgroup size of the run-time-N kernel in fp64, crowded, 6 free-running steps
    against the fp64 oracle."""
    from oracle import formation_oracle as O
    from tests import f64_parity
    rs = np.random.RandomState(N)
    st = O.reset_hd(rs.randint(0, 100000, B), N)
    st["pos"] *= 0.4
    env = f64_parity.Env64(st["pos"], st["vel"], st["ideal_shape"], st["ideal_vel"])
    for t in range(6):
        act = rs.uniform(-1, 1, (B, N, 2)).astype(np.float32).astype(np.float64)
        st, out = O.step_hd(st, act)
        env.step(act)
        np.testing.assert_allclose(env.pos(), st["pos"], rtol=0, atol=1e-10)
        ok = out["cnt_margin"] > 1e-9
        np.testing.assert_allclose(env.indiv.cpu().numpy()[ok], out["indiv"][ok], rtol=0, atol=1e-10)
        np.testing.assert_allclose(env.obs.cpu().numpy(), out["obs"], rtol=0, atol=1e-10)


@pytest.mark.parametrize("name", ["hd_n9", "hd_n9_crowd", "hd_n27", "hd_n27_crowd", "hd_n81", "hd_n81_crowd", "hd_n243"])
def test_f64_pipelined_rollout_kernel_free_runs_on_the_reference_trajectory(golden, name):
    """The PIPELINED rollout kernel (fg_rollout_kernels.hpp: producer / writer waves, double-buffered LDS tables, action prefetch,
    the rows writer) in the fp64 build: the fixture's whole horizon in ONE launch, free-running from the initial state, against
    the reference's float64 trajectory - every step's observations, rewards and done flags and the final state.  The K-step
    path's own link to the reference (it used to inherit the step kernel's); the fp32 product kernels are the same
    source with real = float (byte-identical device code before and after the type was made a parameter)."""
    from tests import f64_parity
    g = golden(name)
    T, B, N = g["acts"].shape[:3]
    r = f64_parity.rollout64(g)
    tol = 1e-9
    assert np.abs(r["pos"] - g["pos"][-1]).max() <= tol and np.abs(r["vel"] - g["vel"][-1]).max() <= tol
    assert (r["step"] == T).all()
    ok = g["cnt_margin"] > 1e-9                                              # [T,B]: no collision count on its threshold
    assert np.abs(r["indiv"] - g["indiv"])[ok].max() <= tol
    shared = g["shared"] if g["shared"].ndim == 3 else np.repeat(g["shared"][..., None], N, -1)
    assert (np.abs(r["reward"] - shared)[ok] <= 1e-9 * np.maximum(1.0, np.abs(shared[ok]))).all()
    np.testing.assert_array_equal(r["done"].astype(bool), g["done"])
    for t in g["obs_steps"]:
        assert np.abs(r["obs"][t - 1] - g["obs_t%d" % t]).max() <= tol, "observation of step %d" % t
    assert np.isfinite(r["obs"]).all() and ok.mean() > 0.9
