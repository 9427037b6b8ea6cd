"""Measured error of the HIP step kernel against the reference's fixtures, shared by tests/test_gpu_f64_parity.py
(asserts the bounds) and profiles/parity_errors.py (writes the committed table profiles/r02_parity_errors.md).

  free_running_f64(g)     the fp64 parity build free-runs over ALL recorded steps from the fixture's initial state
  teacher_forced_f32(g)   the product (fp32) library, state re-seeded from the reference before every step
Both return {quantity: max abs error over all steps / envs / agents} (+ bookkeeping)."""
import numpy as np
import torch

from oracle import formation_oracle as O
from tests import f64_parity

HD_CASES = ["hd_n3", "hd_n4", "hd_n5", "hd_n9", "hd_n10", "hd_n27", "hd_n50", "hd_n81", "hd_n243", "hd_n6_crowd", "hd_n9_crowd", "hd_n16_crowd", "hd_n27_crowd",
            "hd_n81_crowd", "hd_n100_crowd"]


def _mx(a, b, mask=None):
    d = np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))
    if mask is not None:
        d = d[mask]
    return float(d.max()) if d.size else 0.0


def free_running_f64(g, excuse_margin=1e-9):
    T, B, N = g["acts"].shape[:3]
    env = f64_parity.Env64(g["pos0"], g["vel0"], g["ideal_shape"], g["ideal_vel"])
    err = dict(pos=0.0, vel=0.0, indiv=0.0, shared_rel=0.0, obs=0.0)
    idx_bad = dict(near_lm=0, near_ag=0, hd_idx=0, done=0, cnt_excused=0)
    per_step_pos = []
    for t in range(T):
        env.step(g["acts"][t].astype(np.float64))                   # fp32-representable actions, as the reference got them
        err["pos"] = max(err["pos"], _mx(env.pos(), g["pos"][t]))
        err["vel"] = max(err["vel"], _mx(env.vel(), g["vel"][t]))
        per_step_pos.append(_mx(env.pos(), g["pos"][t]))
        ok = g["cnt_margin"][t] > excuse_margin                     # envs where no collision count sits on the threshold
        idx_bad["cnt_excused"] += int((~ok).sum())
        err["indiv"] = max(err["indiv"], _mx(env.indiv.cpu().numpy(), g["indiv"][t], ok))
        sh = env.reward.cpu().numpy()
        rel = np.abs(sh - g["shared"][t]) / np.maximum(1.0, np.abs(g["shared"][t]))
        err["shared_rel"] = max(err["shared_rel"], float(rel[ok].max()) if ok.any() else 0.0)
        idx_bad["done"] += int((env.done.cpu().numpy().astype(bool) != g["done"][t]).sum())
        for k, gap in (("near_lm", "gap_lm"), ("near_ag", "gap_ag")):
            bad = getattr(env, k).cpu().numpy() != g[k][t]
            idx_bad[k] += int((bad & (g[gap][t] > excuse_margin)).sum())
        if (t + 1) in g["obs_steps"]:
            err["obs"] = max(err["obs"], _mx(env.obs.cpu().numpy(), g["obs_t%d" % (t + 1)]))
    return dict(err=err, idx_bad=idx_bad, per_step_pos=per_step_pos, steps=T, envs=B, agents=N)


def teacher_forced_f32(g):
    import formation_gym
    T, B, N = g["acts"].shape[:3]
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
    err = dict(pos=0.0, vel=0.0, indiv=0.0, indiv_same_state=0.0, shared_rel=0.0, obs=0.0, obs_same_state=0.0)
    prev_pos, prev_vel = g["pos0"], g["vel0"]
    f32 = lambda x: np.asarray(x, dtype=np.float32).astype(np.float64)
    for t in range(T):
        env.world.set_state(prev_pos, prev_vel)
        env.scenario.set_formation(env.world, g["ideal_shape"], g["ideal_vel"])
        env.world.step_count.fill_(t)
        obs, rew, done, info = env.step(torch.as_tensor(g["acts"][t]).cuda())
        pos, vel = (x.double().cpu().numpy() for x in env.world.get_state())
        err["pos"] = max(err["pos"], _mx(pos, g["pos"][t]))
        err["vel"] = max(err["vel"], _mx(vel, g["vel"][t]))
        ok = g["cnt_margin"][t] > 1e-5
        ind = info["individual_reward"].double().cpu().numpy()
        err["indiv"] = max(err["indiv"], _mx(ind, g["indiv"][t], ok))
        r = O.reward_hd(pos, vel, f32(g["ideal_shape"]), f32(g["ideal_vel"]), O.HdParams())      # oracle on the GPU's own state
        ok2 = r["cnt_margin"] > 1e-6
        err["indiv_same_state"] = max(err["indiv_same_state"], _mx(ind, r["indiv"], ok2))
        sh = rew[..., 0].double().cpu().numpy()
        rel = np.abs(sh - g["shared"][t]) / np.maximum(1.0, np.abs(g["shared"][t]))
        err["shared_rel"] = max(err["shared_rel"], float(rel[ok].max()) if ok.any() else 0.0)
        if (t + 1) in g["obs_steps"]:
            o = obs.double().cpu().numpy()
            err["obs"] = max(err["obs"], _mx(o, g["obs_t%d" % (t + 1)]))
            err["obs_same_state"] = max(err["obs_same_state"],
                                        _mx(o, O.observation_hd(pos, vel, g["ideal_shape"], g["ideal_vel"])))
        prev_pos, prev_vel = g["pos"][t], g["vel"][t]
    return dict(err=err, steps=T, envs=B, agents=N)


def free_running_f32(g, horizon):
    """The fp32 product free-running from the fixture's initial state: max position error per step."""
    import formation_gym
    T, B, N = g["acts"].shape[:3]
    env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
    env.world.set_state(g["pos0"], g["vel0"])
    env.scenario.set_formation(env.world, g["ideal_shape"], g["ideal_vel"])
    env.world.step_count.zero_()
    out = []
    for t in range(min(T, horizon)):
        env.step(torch.as_tensor(g["acts"][t]).cuda())
        out.append(_mx(env.world.get_state()[0].double().cpu().numpy(), g["pos"][t]))
    return out
