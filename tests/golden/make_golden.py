#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference, which never travels
to the GPU box).  It imports jc-bao/gym-formation's `formation_gym` package
unmodified, drives seeded rollouts through the reference's own
`make_env / env.seed / env.reset / env.step`, and stores inputs + expected
outputs as small .npz files.  Only DATA is stored: no reference source text.

Two import shims are created in a temporary directory at run time (they are not
part of the product and do not touch any arithmetic on the hot path):
  * `gym`   - the reference's environment.py:1-3 imports gym only for the Space
              types (Box/Discrete/Tuple) and EnvSpec; gym is not installed here.
  * `multiagent.{core,scenario}` - basic_formation_env.py:3-4 imports OpenAI
              MPE, which the reference does not vendor; aliased to the
              reference's own formation_gym.core / formation_gym.scenario
              (the only World class that has `world_length`, which
              environment.py:22 reads).

Usage:  python tests/golden/make_golden.py        (rewrites tests/golden/*.npz)
"""
import os
import sys
import tempfile
import textwrap
import warnings

import numpy as np

REF = os.environ.get("FG_REFERENCE", "/root/reference")
OUT = os.environ.get("FG_GOLDEN_OUT") or os.path.dirname(os.path.abspath(__file__))   # FG_GOLDEN_OUT: the drift test writes elsewhere

GYM_SHIM = textwrap.dedent('''
    import numpy as np
    class Env(object):
        pass
    class Space(object):
        pass
    class _Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.shape = tuple(shape) if shape is not None else np.shape(low)
            self.dtype = np.dtype(dtype)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)
        def sample(self):
            return np.random.uniform(self.low, self.high, self.shape).astype(self.dtype)
    class _Discrete(Space):
        def __init__(self, n):
            self.n = n
            self.shape = ()
    class _Tuple(Space):
        def __init__(self, spaces):
            self.spaces = tuple(spaces)
''')


def _install_shims(tmp):
    os.makedirs(os.path.join(tmp, "gym", "envs"))
    with open(os.path.join(tmp, "gym", "__init__.py"), "w") as f:
        f.write(GYM_SHIM + "\nfrom . import spaces\n")
    with open(os.path.join(tmp, "gym", "spaces.py"), "w") as f:
        f.write("from . import _Box as Box, _Discrete as Discrete, _Tuple as Tuple, Space\n")
    with open(os.path.join(tmp, "gym", "envs", "__init__.py"), "w") as f:
        f.write("")
    with open(os.path.join(tmp, "gym", "envs", "registration.py"), "w") as f:
        f.write("class EnvSpec(object):\n    pass\n")
    sys.path.insert(0, tmp)
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True


def _import_reference():
    warnings.simplefilter("ignore")
    import formation_gym  # the reference package
    import formation_gym.core as rcore
    import formation_gym.scenario as rscen
    import types
    ma = types.ModuleType("multiagent")
    sys.modules["multiagent"] = ma
    sys.modules["multiagent.core"] = rcore
    sys.modules["multiagent.scenario"] = rscen
    return formation_gym


def _scenario_of(env):
    # the Scenario instance is reachable through the bound callbacks
    return env.reset_callback.__self__


def _state(env):
    pos = np.array([a.state.p_pos for a in env.world.agents], dtype=np.float64)
    vel = np.array([a.state.p_vel for a in env.world.agents], dtype=np.float64)
    return pos, vel


def _hd_extras(pos, shape, thresh):
    """Derived integer quantities of the reward (reference call site
    formation_hd_env.py:61-75): scipy witness indices, inner argmins, counts."""
    from scipy.spatial.distance import directed_hausdorff
    pt = pos - np.mean(pos, 0)
    d1, i1, j1 = directed_hausdorff(pt, shape)
    d2, i2, j2 = directed_hausdorff(shape, pt)
    D = np.linalg.norm(pt[:, None, :] - shape[None, :, :], axis=2)
    near_lm = np.argmin(D, axis=1)       # per agent: nearest ideal point
    near_ag = np.argmin(D, axis=0)       # per ideal point: nearest agent
    PD = np.linalg.norm(pos[:, None, :] - pos[None, :, :], axis=2)
    cnt = (PD < thresh).sum(1) - 1       # j != i (PD[i,i] = 0 < thresh)
    # top-2 gap of each argmin (to let fp32 tests excuse genuine near-ties)
    Ds = np.sort(D, axis=1)
    gap_lm = Ds[:, 1] - Ds[:, 0]
    Ds0 = np.sort(D, axis=0)
    gap_ag = Ds0[1, :] - Ds0[0, :]
    # smallest |dist - thresh| over pairs (near-tie excuse for counts)
    off = np.abs(PD - thresh) + np.eye(len(pos))
    return dict(hd=np.array([d1, d2]), hd_idx=np.array([i1, j1, i2, j2], dtype=np.int32),
                near_lm=near_lm.astype(np.int32), near_ag=near_ag.astype(np.int32),
                cnt=cnt.astype(np.int32), gap_lm=gap_lm, gap_ag=gap_ag,
                cnt_margin=np.array(off.min()))


WALLS = [("V", -0.9, (-1.0, 1.0), 0.1), ("V", 0.9, (-0.6, 0.6), 0.1), ("H", 0.8, (-0.5, 0.5), 0.2)]
SOFT_WALLS = [("H", -0.05, (-0.4, 0.4), 0.1)]            # through the crowd: felt by everybody but the ghosts


def rollout_hd(fg, N, B, T, seed, act_seed, crowd=None, obs_at=None, options=None):
    """Seeded rollout of formation_hd_env through the reference API.  `options` switches on
    the World features no reference scenario enables: max_speed (core.py:271-276), accel
    (core.py:236, environment.py:219-220), walls (core.py:255-261,325-362)."""
    obs_at = set(obs_at or [1, T])
    acts = np.random.RandomState(act_seed).uniform(-1, 1, (T, B, N, 2)).astype(np.float32)
    out = {k: [] for k in ("pos", "vel", "indiv", "shared", "done", "hd", "hd_idx",
                           "near_lm", "near_ag", "cnt", "gap_lm", "gap_ag", "cnt_margin")}
    obs_store = {t: [] for t in sorted(obs_at)}
    pos0, vel0, shp, ivel, obs0 = [], [], [], [], []
    for b in range(B):
        env = fg.make_env("formation_hd_env", False, N)
        if options:
            import formation_gym.core as rcore
            for a in env.world.agents:
                a.max_speed = options.get("max_speed")
                a.accel = options.get("accel")
            if options.get("walls"):
                env.world.walls = [rcore.Wall(o, ax, ep, w) for (o, ax, ep, w) in WALLS]
            if options.get("soft_walls"):                 # a ghost entity passes through a soft wall (core.py:326-327)
                env.world.walls = env.world.walls + [rcore.Wall(o, ax, ep, w, hard=False) for (o, ax, ep, w) in SOFT_WALLS]
            fl = options.get("flags")                     # Entity.collide / Entity.ghost per agent (core.py:54-58)
            if fl:
                for a, c_, g_ in zip(env.world.agents, fl["collide"], fl["ghost"]):
                    a.collide = bool(c_); a.ghost = bool(g_)
            het = options.get("hetero")                   # per-agent mass / size / accel / max_speed (core.py:45-109)
            if het:
                for a, m_, s_, ac_, ms_ in zip(env.world.agents, het["mass"], het["size"], het["accel"], het["max_speed"]):
                    a.initial_mass = float(m_); a.size = float(s_)
                    a.accel = None if np.isnan(ac_) else float(ac_)
                    a.max_speed = None if np.isnan(ms_) else float(ms_)
            wc = options.get("world")                     # non-default World constants (core.py:119-139)
            if wc:
                env.world.dt = wc["dt"]; env.world.damping = wc["damping"]
                env.world.contact_force = wc["contact_force"]; env.world.contact_margin = wc["contact_margin"]
                env.world.world_length = wc["world_length"]; env.world_length = wc["world_length"]   # environment.py:22
                for a in env.world.agents:
                    a.initial_mass = wc["mass"]; a.size = wc["size"]
        env.seed(seed + 1000 * b)
        o0 = env.reset()
        sc = _scenario_of(env)
        if crowd is not None:
            for a in env.world.agents:
                a.state.p_pos = a.state.p_pos * crowd
            o0 = [env._get_obs(a) for a in env.agents]
        p, v = _state(env)
        pos0.append(p); vel0.append(v)
        shp.append(np.array(sc.ideal_shape, dtype=np.float64))
        ivel.append(np.array(sc.ideal_vel, dtype=np.float64))
        obs0.append(np.array(o0, dtype=np.float64))
        thresh = (env.world.agents[0].size + env.world.agents[1].size) / 2
        if options and options.get("hetero"):             # is_collision per pair (formation_hd_env.py:119-121)
            sz_ = np.array([a.size for a in env.world.agents])
            thresh = (sz_[:, None] + sz_[None, :]) / 2
        rec = {k: [] for k in out}
        for t in range(T):
            act_n = [acts[t, b, i].astype(np.float64).copy() for i in range(N)]
            obs_n, rew_n, done_n, info_n = env.step(act_n)
            p, v = _state(env)
            rec["pos"].append(p); rec["vel"].append(v)
            rec["indiv"].append(np.array([inf["individual_reward"] for inf in info_n]))
            rec["shared"].append(np.array([r[0] for r in rew_n]))
            rec["done"].append(np.array(done_n, dtype=np.bool_))
            ex = _hd_extras(p, np.array(sc.ideal_shape), thresh)
            for k, val in ex.items():
                rec[k].append(val)
            if (t + 1) in obs_at:
                obs_store[t + 1].append(np.array(obs_n, dtype=np.float64))
        for k in out:
            out[k].append(np.array(rec[k]))
    res = {k: np.stack(v, axis=1) for k, v in out.items()}   # [T,B,...]
    res.update(acts=acts, pos0=np.array(pos0), vel0=np.array(vel0),
               ideal_shape=np.array(shp), ideal_vel=np.array(ivel), obs0=np.array(obs0),
               seed=np.array(seed), act_seed=np.array(act_seed),
               crowd=np.array(-1.0 if crowd is None else crowd),
               obs_steps=np.array(sorted(obs_at), dtype=np.int32))
    for t, lst in obs_store.items():
        res["obs_t%d" % t] = np.array(lst)                     # [B,N,6N]
    if options and options.get("world"):
        res.update({"world_" + k: np.array(v) for k, v in options["world"].items()})
    if options and options.get("hetero"):
        res.update({"agent_" + k: np.array(v, dtype=np.float64) for k, v in options["hetero"].items()})
    if options and options.get("flags"):
        res.update({"agent_" + k: np.array(v, dtype=np.bool_) for k, v in options["flags"].items()})
    return res


def immovable_fixture(fg, N, T, seed, act_seed, crowd):
    """An immovable agent (core.py:231, 266-267, 294-295, 319-321) and a non-colliding one (:292-293) among agents of
    different mass.  The reference's env.step cannot take a silent immovable agent (`assert len(action) == 0`,
    environment.py:236 - recorded below), so the World is driven through core.py's own API as in comm_fixture."""
    env = fg.make_env("formation_hd_env", False, N)
    sc = _scenario_of(env)
    world = env.world
    movable = np.ones(N, dtype=bool); movable[1] = False
    collide = np.ones(N, dtype=bool); collide[N - 2] = False
    mass = np.random.RandomState(seed).uniform(0.5, 2.5, N)
    for a, m_, c_, ms_ in zip(world.agents, movable, collide, mass):
        a.movable = bool(m_); a.collide = bool(c_); a.initial_mass = float(ms_)
    env.seed(seed)
    env.reset()
    for a in world.agents:
        a.state.p_pos = a.state.p_pos * crowd
    world.agents[1].state.p_vel = np.array([0.3, -0.2])          # an immovable agent keeps whatever velocity it has (:266-267)
    p0, v0 = _state(env)
    try:
        env.step([np.zeros(2) for _ in range(N)])
        raised = "none"
    except Exception as exc:                                      # noqa: BLE001
        raised = type(exc).__name__ + ": " + str(exc)
    for a, p_, v_ in zip(world.agents, p0, v0):                  # undo whatever the failed call did
        a.state.p_pos = p_.copy(); a.state.p_vel = v_.copy()
    acts = np.random.RandomState(act_seed).uniform(-1, 1, (T, N, 2)).astype(np.float32)
    rec = {k: [] for k in ("pos", "vel", "obs", "indiv")}
    for t in range(T):
        for i, a in enumerate(world.agents):
            a.action.u = 5.0 * acts[t, i].astype(np.float64)     # what _set_action does for a movable agent (environment.py:216-221)
        world.step()
        p, v = _state(env)
        rec["pos"].append(p); rec["vel"].append(v)
        rec["obs"].append(np.array([sc.observation(a, world) for a in world.agents], dtype=np.float64))
        rec["indiv"].append(np.array([sc.reward(a, world) for a in world.agents], dtype=np.float64))
    res = {k: np.array(v) for k, v in rec.items()}
    res.update(pos0=p0, vel0=v0, acts=acts, movable=movable, collide=collide, mass=mass,
               ideal_shape=np.array(sc.ideal_shape, dtype=np.float64), ideal_vel=np.array(sc.ideal_vel, dtype=np.float64),
               seed=np.array(seed), env_step_raises=np.array(raised))
    return res


def scripted_u(p_own, v_own, p_lead):
    """The deterministic scripted agent of fixture hd_n6_scripted: orbit the origin, damp, lean towards agent 0.  Written
    with + - * only, on the last axis, so that tests/test_gpu_hetero_comm.py can evaluate the SAME function on [B, 2] tensors."""
    return 0.6 * np.stack([-p_own[..., 1], p_own[..., 0]], -1) - 0.3 * v_own + 0.2 * (p_lead - p_own)


def scripted_fixture(fg, N, T, seed, act_seed, crowd):
    """Scripted agents (Agent.action_callback, core.py:152-158, 210-211): World.step calls the callback and uses its action.u
    as it is; the other agents get 5 x the raw action as _set_action would give them.  Driven through core.py's own API."""
    core = sys.modules[type(fg.make_env("formation_hd_env", False, 3).world).__module__]
    env = fg.make_env("formation_hd_env", False, N)
    sc = _scenario_of(env)
    world = env.world
    scripted = np.zeros(N, dtype=bool); scripted[[2, N - 1]] = True
    mass = np.random.RandomState(seed).uniform(0.5, 2.0, N)

    def callback(agent, w):
        act = core.Action()
        act.u = scripted_u(agent.state.p_pos, agent.state.p_vel, w.agents[0].state.p_pos)
        act.c = np.zeros(w.dim_c)
        return act
    for a, s_, m_ in zip(world.agents, scripted, mass):
        a.initial_mass = float(m_)
        if s_:
            a.action_callback = callback
    assert len(world.scripted_agents) == 2 and len(world.policy_agents) == N - 2
    env.seed(seed)
    env.reset()
    for a in world.agents:
        a.state.p_pos = a.state.p_pos * crowd
    p0, v0 = _state(env)
    acts = np.random.RandomState(act_seed).uniform(-1, 1, (T, N, 2)).astype(np.float32)
    rec = {k: [] for k in ("pos", "vel", "obs", "indiv", "u_scripted")}
    for t in range(T):
        for i, a in enumerate(world.agents):
            if not scripted[i]:
                a.action.u = 5.0 * acts[t, i].astype(np.float64)     # what _set_action does for a policy agent (environment.py:216-221)
        world.step()
        p, v = _state(env)
        rec["pos"].append(p); rec["vel"].append(v)
        rec["u_scripted"].append(np.array([a.action.u for a in world.scripted_agents], dtype=np.float64))
        rec["obs"].append(np.array([sc.observation(a, world) for a in world.agents], dtype=np.float64))
        rec["indiv"].append(np.array([sc.reward(a, world) for a in world.agents], dtype=np.float64))
    res = {k: np.array(v) for k, v in rec.items()}
    res.update(pos0=p0, vel0=v0, acts=acts, scripted=scripted, mass=mass,
               ideal_shape=np.array(sc.ideal_shape, dtype=np.float64), ideal_vel=np.array(sc.ideal_vel, dtype=np.float64),
               seed=np.array(seed))
    return res


def hetero_options(N, seed):
    """Per-agent properties no reference scenario sets: mass 0.5 ... 3, size 0.02 ... 0.09, a third of the agents with
    their own accel, a third with a max_speed (NaN = None)."""
    rs = np.random.RandomState(seed)
    mass = rs.uniform(0.5, 3.0, N); size = rs.uniform(0.02, 0.09, N)
    accel = np.where(rs.uniform(size=N) < 0.34, rs.uniform(2.0, 6.0, N), np.nan)
    max_speed = np.where(rs.uniform(size=N) < 0.34, rs.uniform(0.2, 0.8, N), np.nan)
    return dict(mass=mass, size=size, accel=accel, max_speed=max_speed)


def comm_fixture(fg, N, T, seed, act_seed, crowd):
    """Non-silent agents (core.py:279-286, formation_hd_env.py:48-51).  The reference's env.step cannot take them
    (environment.py:231 indexes an exhausted action list -> IndexError, recorded below), so the World is driven the
    way core.py's own API allows: action.u / action.c set per agent, world.step(), then the scenario callbacks."""
    env = fg.make_env("formation_hd_env", False, N)
    sc = _scenario_of(env)
    world = env.world
    silent = np.zeros(N, dtype=bool); silent[1] = True           # one agent stays silent: its c is zeros
    for a, s_ in zip(world.agents, silent):
        a.silent = bool(s_)
    env.seed(seed)
    env.reset()
    for a in world.agents:
        a.state.p_pos = a.state.p_pos * crowd
    p0, v0 = _state(env)
    try:
        env.step([np.zeros(2) for _ in range(N)])
        raised = "none"
    except Exception as exc:                                      # noqa: BLE001
        raised = type(exc).__name__ + ": " + str(exc)
    for a, p_, v_ in zip(world.agents, p0, v0):                  # undo whatever the failed call did
        a.state.p_pos = p_.copy(); a.state.p_vel = v_.copy()
    rs = np.random.RandomState(act_seed)
    acts = rs.uniform(-1, 1, (T, N, 2)).astype(np.float32)
    comm = rs.uniform(0, 1, (T, N, 2)).astype(np.float32)
    rec = {k: [] for k in ("pos", "vel", "c", "obs", "indiv")}
    for t in range(T):
        for i, a in enumerate(world.agents):
            a.action.u = 5.0 * acts[t, i].astype(np.float64)     # what _set_action does (environment.py:216-221)
            a.action.c = comm[t, i].astype(np.float64)
        world.step()
        p, v = _state(env)
        rec["pos"].append(p); rec["vel"].append(v)
        rec["c"].append(np.array([a.state.c for a in world.agents], dtype=np.float64))
        rec["obs"].append(np.array([sc.observation(a, world) for a in world.agents], dtype=np.float64))
        rec["indiv"].append(np.array([sc.reward(a, world) for a in world.agents], dtype=np.float64))
    res = {k: np.array(v) for k, v in rec.items()}
    res.update(pos0=p0, vel0=v0, acts=acts, comm=comm, silent=silent, ideal_shape=np.array(sc.ideal_shape, dtype=np.float64),
               ideal_vel=np.array(sc.ideal_vel, dtype=np.float64), seed=np.array(seed), env_step_raises=np.array(raised))
    return res


def rollout_action_mode(fg, mode, N, T, seed, act_seed):
    """formation_hd_env driven through the non-default branches of _set_action
    (environment.py:187-216), which make_env never selects: `onehot5` = MultiAgentEnv(...,
    discrete_action=True), `index` = discrete_action_input, `argmax` = world.discrete_action."""
    rs = np.random.RandomState(act_seed)
    base = fg.make_env("formation_hd_env", False, N)
    sc = _scenario_of(base)
    world = base.world
    if mode == "argmax":
        world.discrete_action = True
    env = fg.MultiAgentEnv(world, sc.reset_world, sc.reward, sc.observation, shared_viewer=True,
                           discrete_action=(mode == "onehot5"))
    if mode == "index":
        env.discrete_action_input = True
    if mode == "onehot5":
        acts = rs.uniform(0, 1, (T, N, 5))
    elif mode == "index":
        acts = rs.randint(0, 5, (T, N)).astype(np.int32)
    else:
        acts = rs.uniform(-1, 1, (T, N, 2))
    env.seed(seed)
    obs0 = np.array(env.reset(), dtype=np.float64)
    pos0, vel0 = _state(env)
    rec = {k: [] for k in ("pos", "vel", "indiv", "shared", "obs", "acts_after")}
    for t in range(T):
        if mode == "index":
            act_n = [int(acts[t, i]) for i in range(N)]
        else:
            act_n = [acts[t, i].astype(np.float64).copy() for i in range(N)]
        obs_n, rew_n, done_n, info_n = env.step(act_n)
        p, v = _state(env)
        rec["pos"].append(p); rec["vel"].append(v)
        rec["indiv"].append(np.array([inf["individual_reward"] for inf in info_n]))
        rec["shared"].append(np.array([r[0] for r in rew_n]))
        rec["obs"].append(np.array(obs_n, dtype=np.float64))
        rec["acts_after"].append(np.array(act_n, dtype=np.float64))      # what the call left in the caller's arrays
    res = {k: np.array(v) for k, v in rec.items()}
    res.update(acts=acts, pos0=pos0, vel0=vel0, obs0=obs0, ideal_shape=np.array(sc.ideal_shape, dtype=np.float64),
               ideal_vel=np.array(sc.ideal_vel, dtype=np.float64), seed=np.array(seed),
               action_space_n=np.array(getattr(env.action_space[0], "n", -1)))
    return res


def _scn_world_options(env, hetero=None, flags=None, walls=False):
    """hetero: per-agent mass / size / max_speed (core.py:45-109); flags: Entity.movable / collide / ghost per agent
    (core.py:54-58); walls: WALLS + SOFT_WALLS (core.py:255-261, 325-362) - all on top of the scenario's own make_world."""
    import formation_gym.core as rcore
    if hetero:
        for a, m_, s_, ms_ in zip(env.world.agents, hetero["mass"], hetero["size"], hetero["max_speed"]):
            a.initial_mass = float(m_); a.size = float(s_)
            a.max_speed = None if np.isnan(ms_) else float(ms_)
    if flags:
        for i, a in enumerate(env.world.agents):
            a.collide = bool(flags["collide"][i]); a.ghost = bool(flags["ghost"][i])
            if "movable" in flags:
                a.movable = bool(flags["movable"][i])
    if walls:
        env.world.walls = ([rcore.Wall(o, ax, ep, w) for (o, ax, ep, w) in WALLS] +
                           [rcore.Wall(o, ax, ep, w, hard=False) for (o, ax, ep, w) in SOFT_WALLS])


def rollout_scn(fg, name, N, B, T, seed, act_seed, crowd=None, hetero=None, flags=None, walls=False):
    """Seeded rollout of one of the landmark scenarios (basic_formation_env, formation_hd_partial_env,
    formation_hd_partial_range_env, formation_hd_obs_env) through the reference API.  hetero: per-agent mass / size /
    max_speed (core.py:45-109) on top of the file's own make_world - the obstacles keep theirs; flags: agents that do not
    collide, ghosts (core.py:54-58); walls: WALLS + SOFT_WALLS."""
    acts = np.random.RandomState(act_seed).uniform(-1, 1, (T, B, N, 2)).astype(np.float32)
    keys = ("pos", "vel", "lm", "lmvel", "obs", "indiv", "shared", "done")
    out = {k: [] for k in keys}
    init = {k: [] for k in ("pos0", "vel0", "lm0", "lmvel0", "obs0")}
    meta = {}
    for b in range(B):
        env = fg.make_env(name, False, N)
        _scn_world_options(env, hetero, flags, walls)
        env.seed(seed + 1000 * b)
        o0 = env.reset()
        if crowd is not None:
            for a in env.world.agents:
                a.state.p_pos = a.state.p_pos * crowd
            o0 = [env._get_obs(a) for a in env.agents]
        sc = _scenario_of(env)
        lm = lambda: np.array([l.state.p_pos for l in env.world.landmarks], dtype=np.float64)
        lmv = lambda: np.array([l.state.p_vel for l in env.world.landmarks], dtype=np.float64)
        p, v = _state(env)
        init["pos0"].append(p); init["vel0"].append(v); init["lm0"].append(lm()); init["lmvel0"].append(lmv())
        init["obs0"].append(np.array(o0, dtype=np.float64))
        rec = {k: [] for k in keys}
        for t in range(T):
            act_n = [acts[t, b, i].astype(np.float64).copy() for i in range(N)]
            obs_n, rew_n, done_n, info_n = env.step(act_n)
            p, v = _state(env)
            rec["pos"].append(p); rec["vel"].append(v); rec["lm"].append(lm()); rec["lmvel"].append(lmv())
            rec["obs"].append(np.array(obs_n, dtype=np.float64))
            rec["indiv"].append(np.array([inf["individual_reward"] for inf in info_n]))
            rec["shared"].append(np.array([r[0] for r in rew_n]))
            rec["done"].append(np.array(done_n, dtype=np.bool_))
        for k in keys:
            out[k].append(np.array(rec[k]))
        meta = dict(world_length=np.array(env.world_length), agent_size=np.array(env.world.agents[0].size),
                    num_landmarks=np.array(getattr(sc, "num_landmarks", len(env.world.landmarks))),
                    num_entities=np.array(len(env.world.landmarks)),
                    num_obs=np.array(getattr(sc, "num_obs", -1)), obs_range=np.array(getattr(sc, "obs_range", -1.0)),
                    lm_sizes=np.array([l.size for l in env.world.landmarks]),
                    obs_dim=np.array(env.observation_space[0].shape[0]))
    res = {k: np.stack(v, axis=1) for k, v in out.items()}
    res.update({k: np.array(v) for k, v in init.items()})
    res.update(meta)
    res.update(acts=acts, seed=np.array(seed), act_seed=np.array(act_seed),
               crowd=np.array(-1.0 if crowd is None else crowd))
    if hetero:
        res.update({"agent_" + k: np.array(v, dtype=np.float64) for k, v in hetero.items()})
    if flags:
        res.update({"agent_" + k: np.array(v, dtype=np.bool_) for k, v in flags.items()})
    return res


def scn_immovable_fixture(fg, name, N, T, seed, act_seed, crowd, hetero, flags):
    """An immovable agent (core.py:231, 266-267, 294-295, 319-321) in a landmark scenario, next to non-colliding agents, ghosts,
    walls and (formation_hd_obs_env) the falling obstacles, which are pushed by the immovable agent with the plain force
    (:319-321).  env.step cannot take a silent immovable agent (environment.py:236, fixture hd_n6_immovable), so the World is
    driven through core.py's own API: action.u = 5 * action (environment.py:216-221), world.step(), then the scenario's
    observation and reward callbacks per agent in env.step's order (the obstacle scenario's reward re-arms the obstacle
    velocities, formation_hd_obs_env.py:84-89)."""
    env = fg.make_env(name, False, N)
    _scn_world_options(env, hetero, flags, walls=True)
    sc = _scenario_of(env)
    world = env.world
    env.seed(seed)
    env.reset()
    for a in world.agents:
        a.state.p_pos = a.state.p_pos * crowd
    for i, a in enumerate(world.agents):
        if not a.movable:
            a.state.p_vel = np.array([0.25, -0.15])               # an immovable agent keeps whatever velocity it has
    if name == "formation_hd_obs_env":                            # bring the obstacles down to the agents
        for k, l in enumerate(world.landmarks[sc.num_landmarks:]):
            l.state.p_pos = np.array([0.3 * (k - 1), 0.45 + 0.05 * k])
    lm = lambda: np.array([l.state.p_pos for l in world.landmarks], dtype=np.float64)
    lmv = lambda: np.array([l.state.p_vel for l in world.landmarks], dtype=np.float64)
    p0, v0 = _state(env)
    init = dict(pos0=p0[None], vel0=v0[None], lm0=lm()[None], lmvel0=lmv()[None])
    acts = np.random.RandomState(act_seed).uniform(-1, 1, (T, 1, N, 2)).astype(np.float32)
    keys = ("pos", "vel", "lm", "lmvel", "obs", "indiv")
    rec = {k: [] for k in keys}
    for t in range(T):
        for i, a in enumerate(world.agents):
            a.action.u = 5.0 * acts[t, 0, i].astype(np.float64)
        world.step()
        obs_n, rew_n = [], []
        for a in world.agents:
            obs_n.append(sc.observation(a, world)); rew_n.append(sc.reward(a, world))
        p, v = _state(env)
        rec["pos"].append(p[None]); rec["vel"].append(v[None]); rec["lm"].append(lm()[None]); rec["lmvel"].append(lmv()[None])
        rec["obs"].append(np.array(obs_n, dtype=np.float64)[None]); rec["indiv"].append(np.array(rew_n, dtype=np.float64)[None])
    res = {k: np.array(v) for k, v in rec.items()}                # [T,1,...]
    res.update(init)
    res.update(acts=acts, seed=np.array(seed), act_seed=np.array(act_seed), crowd=np.array(crowd),
               num_landmarks=np.array(getattr(sc, "num_landmarks", len(world.landmarks))))
    res.update({"agent_" + k: np.array(v, dtype=np.float64) for k, v in hetero.items()})
    res.update({"agent_" + k: np.array(v, dtype=np.bool_) for k, v in flags.items()})
    return res


def rollout_basic(fg, N, T, seed, act_seed):
    acts = np.random.RandomState(act_seed).uniform(-1, 1, (T, 1, N, 2)).astype(np.float32)
    env = fg.make_env("basic_formation_env", False, N)
    env.seed(seed)
    o0 = env.reset()
    L = len(env.world.landmarks)
    lm = np.array([l.state.p_pos for l in env.world.landmarks])
    p0, v0 = _state(env)
    rec = {k: [] for k in ("pos", "vel", "indiv", "shared", "done", "obs")}
    for t in range(T):
        act_n = [acts[t, 0, i].astype(np.float64).copy() for i in range(N)]
        obs_n, rew_n, done_n, info_n = env.step(act_n)
        p, v = _state(env)
        rec["pos"].append(p); rec["vel"].append(v)
        rec["indiv"].append(np.array([inf["individual_reward"] for inf in info_n]))
        rec["shared"].append(np.array([r[0] for r in rew_n]))
        rec["done"].append(np.array(done_n, dtype=np.bool_))
        rec["obs"].append(np.array(obs_n, dtype=np.float64))
    res = {k: np.array(v)[:, None] for k, v in rec.items()}    # [T,1,...]
    res.update(acts=acts, pos0=p0[None], vel0=v0[None], landmarks=lm[None],
               obs0=np.array(o0)[None], seed=np.array(seed), act_seed=np.array(act_seed),
               world_length=np.array(env.world_length), num_landmarks=np.array(L),
               agent_size=np.array(env.world.agents[0].size),
               obs_dim=np.array(env.observation_space[0].shape[0]),
               share_obs_dim=np.array(env.share_observation_space[0].shape[0]))
    return res


def reset_fixture(fg, cases):
    res = {}
    for (seed, N) in cases:
        env = fg.make_env("formation_hd_env", False, N)
        env.seed(seed)
        o0 = env.reset()
        sc = _scenario_of(env)
        p, v = _state(env)
        lm = np.array([l.state.p_pos for l in env.world.landmarks])
        key = "s%d_n%d" % (seed, N)
        res[key + "_pos"] = p
        res[key + "_vel"] = v
        res[key + "_shape"] = np.array(sc.ideal_shape)
        res[key + "_ivel"] = np.array(sc.ideal_vel)
        res[key + "_landmarks"] = lm          # after the observation side effect (:40-44)
        res[key + "_obs"] = np.array(o0)
        res[key + "_obs_dim"] = np.array(env.observation_space[0].shape[0])
        res[key + "_share_obs_dim"] = np.array(env.share_observation_space[0].shape[0])
        res[key + "_world_length"] = np.array(env.world_length)
    res["cases"] = np.array(cases, dtype=np.int64)
    return res


def shape_fixture(fg):
    env = fg.make_env("formation_hd_env", False, 3)
    sc = _scenario_of(env)
    res = {}
    for L in range(4):
        res["layer%d" % L] = np.array(sc.generate_shape(L), dtype=np.float64).reshape(-1, 2)
    return res


def policy_fixture(fg, N, T, seed, per=3):
    """ezpolicy / get_action_BFS driven closed loop (reference __init__.py:19-99)."""
    env = fg.make_env("formation_hd_env", False, N)
    env.seed(seed)
    obs_n = env.reset()
    sc = _scenario_of(env)
    p0, v0 = _state(env)
    rec = {k: [] for k in ("act", "pos", "vel", "shared")}
    obs_first = np.array(obs_n)
    for t in range(T):
        act_n = fg.get_action_BFS(fg.ezpolicy, obs_n, per)
        rec["act"].append(np.array(act_n, dtype=np.float64))   # before env.step scales it in place
        obs_n, rew_n, done_n, _ = env.step([np.array(a, dtype=np.float64) for a in act_n])
        p, v = _state(env)
        rec["pos"].append(p); rec["vel"].append(v)
        rec["shared"].append(rew_n[0][0])
    res = {k: np.array(v) for k, v in rec.items()}
    res.update(pos0=p0, vel0=v0, ideal_shape=np.array(sc.ideal_shape),
               ideal_vel=np.array(sc.ideal_vel), obs0=obs_first, seed=np.array(seed), per=np.array(per))
    # single-agent-view ezpolicy known answers on the initial observations
    res["ez_act0"] = np.array([fg.ezpolicy(o) for o in obs_first]) if N == 3 else np.zeros(0)
    return res


def benchmark_fixture(fg, N, T, seed, act_seed, crowd):
    """make_env(..., benchmark=True): per step the reference's step outputs (unchanged by the flag:
    environment.py:130-133 forwards only a 'fail' key, which benchmark_data never sets) and, per agent,
    the dict `Scenario.benchmark_data` (formation_hd_env.py:97-117) returns through `env._get_info`."""
    env = fg.make_env("formation_hd_env", True, N)
    env.seed(seed)
    env.reset()
    for a in env.world.agents:                              # crowd the agents so that collisions occur
        a.state.p_pos = a.state.p_pos * crowd
    sc = _scenario_of(env)
    p0, v0 = _state(env)
    lm0 = np.array([l.state.p_pos for l in env.world.landmarks], dtype=np.float64)
    acts = np.random.RandomState(act_seed).uniform(-1, 1, (T, N, 2)).astype(np.float32)
    rec = {k: [] for k in ("pos", "vel", "shared", "indiv", "info_keys", "b_reward", "b_collisions", "b_min_dists",
                           "b_occupied", "lm")}
    for t in range(T):
        obs_n, rew_n, done_n, info_n = env.step([acts[t, i].astype(np.float64) for i in range(N)])
        p, v = _state(env)
        rec["pos"].append(p); rec["vel"].append(v)
        rec["shared"].append(rew_n[0][0])
        rec["indiv"].append([i["individual_reward"] for i in info_n])
        rec["info_keys"].append(sorted(set(k for i in info_n for k in i.keys())) == ["individual_reward"])
        infos = [env._get_info(a) for a in env.world.agents]
        rec["b_reward"].append([i["reward"] for i in infos])
        rec["b_collisions"].append([i["collisions"] for i in infos])
        rec["b_min_dists"].append([i["min_dists"] for i in infos])
        rec["b_occupied"].append([i["occupied_landmarks"] for i in infos])
        rec["lm"].append(np.array([l.state.p_pos for l in env.world.landmarks], dtype=np.float64))
    res = {k: np.array(v) for k, v in rec.items()}
    res.update(pos0=p0, vel0=v0, lm0=lm0, acts=acts, ideal_shape=np.array(sc.ideal_shape),
               ideal_vel=np.array(sc.ideal_vel), seed=np.array(seed))
    return res


def plugin_fixture(fg, path, N, T, seed, act_seed):
    """An ORIGINAL reference-style scenario file (tests/plugins/) run under the REAL reference: loaded the way
    make_env loads envs/<name>.py (__init__.py:8-16, which only accepts names inside the reference's own package)."""
    import imp
    scenario = imp.load_source('', path).Scenario()
    world = scenario.make_world(N)
    env = fg.MultiAgentEnv(world, scenario.reset_world, scenario.reward, scenario.observation, shared_viewer=True)
    env.seed(seed)
    obs0 = np.array(env.reset(), dtype=np.float64)
    p0, v0 = _state(env)
    acts = np.random.RandomState(act_seed).uniform(-1, 1, (T, N, 2)).astype(np.float32)
    rec = {k: [] for k in ("pos", "vel", "obs", "indiv", "shared", "done", "lm_pos", "lm_vel")}
    lm0 = np.array([l.state.p_pos for l in world.landmarks], dtype=np.float64)
    lv0 = np.array([l.state.p_vel if l.state.p_vel is not None else np.zeros(2) for l in world.landmarks], dtype=np.float64)
    for t in range(T):
        obs_n, rew_n, done_n, info_n = env.step([acts[t, i].astype(np.float64) for i in range(N)])
        p, v = _state(env)
        rec["pos"].append(p); rec["vel"].append(v)
        rec["lm_pos"].append(np.array([l.state.p_pos for l in world.landmarks], dtype=np.float64))
        rec["lm_vel"].append(np.array([l.state.p_vel if l.state.p_vel is not None else np.zeros(2) for l in world.landmarks],
                                      dtype=np.float64))
        rec["obs"].append(np.array(obs_n, dtype=np.float64))
        rec["indiv"].append(np.array([inf["individual_reward"] for inf in info_n]))
        rec["shared"].append(rew_n[0][0])
        rec["done"].append(np.array(done_n, dtype=np.bool_))
    res = {k: np.array(v) for k, v in rec.items()}
    res.update(pos0=p0, vel0=v0, obs0=obs0, acts=acts, seed=np.array(seed), radius=np.array(getattr(scenario, "radius", 0.0)),
               lm_pos0=lm0, lm_vel0=lv0,
               beacon=np.array(world.landmarks[0].state.p_pos, dtype=np.float64),
               obs_dim=np.array(env.observation_space[0].shape[0]), world_length=np.array(env.world_length))
    return res


def vec_env_fixture(fg, N, B, T, seed, act_seed):
    """What an RL caller of the reference gets from its vectorised env (train/maddpg-v2/utils/env_wrappers.py): B envs seeded
    seed + 1000 rank (main.py:19-30), stepped by the body of DummyVecEnv.step_wait (:113-122; `baselines`, its base class, is
    absent, so the loop is restated HERE, in the generator): stack, ts += 1, reset the envs whose agents are all done, return
    the RESET observation with the finished step's rewards / dones.  SubprocVecEnv's worker does the same per env (:14-18)."""
    envs = []
    for rank in range(B):
        env = fg.make_env("formation_hd_env", False, N)
        env.seed(seed + rank * 1000)
        np.random.seed(seed + rank * 1000)                       # main.py:24-25
        envs.append(env)
    acts = np.random.RandomState(act_seed).uniform(-1, 1, (T, B, N, 2)).astype(np.float32)
    states = [np.random.RandomState(seed + rank * 1000).get_state() for rank in range(B)]

    def with_stream(b, fn):                                       # every env draws from its own global stream, as its process would
        saved = np.random.get_state()
        np.random.set_state(states[b])
        try:
            return fn()
        finally:
            states[b] = np.random.get_state()
            np.random.set_state(saved)
    obs0 = np.array([with_stream(b, envs[b].reset) for b in range(B)])
    ts = np.zeros(B, dtype='int')
    rec = {k: [] for k in ("obs", "rews", "dones", "ts", "indiv")}
    info_kind = None
    for t in range(T):
        results = [env.step([a_.astype(np.float64) for a_ in a]) for (a, env) in zip(acts[t], envs)]
        obs, rews, dones, infos = map(np.array, zip(*results))
        ts += 1
        for (i, done) in enumerate(dones):
            if all(done):
                obs[i] = with_stream(i, envs[i].reset)
                ts[i] = 0
        rec["obs"].append(np.array(obs)); rec["rews"].append(np.array(rews)); rec["dones"].append(np.array(dones))
        rec["ts"].append(ts.copy())
        rec["indiv"].append(np.array([[d["individual_reward"] for d in row] for row in infos]))
        info_kind = "%s %s %s" % (type(infos).__name__, infos.shape, sorted(infos[0][0].keys()))
    res = {k: np.array(v) for k, v in rec.items()}
    res.update(obs0=obs0, acts=acts, seed=np.array(seed), info_kind=np.array(info_kind),
               agent_types=np.array(['adversary' if a.adversary else 'agent' for a in envs[0].agents]),
               world_length=np.array(envs[0].world_length))
    return res


def hausdorff_kat():
    """scipy's own published docstring example for directed_hausdorff
    (scipy 1.15.3 spatial/distance.py) - the only external KAT on this path."""
    from scipy.spatial.distance import directed_hausdorff
    u = np.array([(1.0, 0.0), (0.0, 1.0), (-1.0, 0.0), (0.0, -1.0)])
    v = np.array([(2.0, 0.0), (0.0, 2.0), (-2.0, 0.0), (0.0, -4.0)])
    return dict(u=u, v=v, d_uv=np.array(directed_hausdorff(u, v)[0]),
                d_vu=np.array(directed_hausdorff(v, u)[0]))


def main():
    only = sys.argv[1:]          # optional: regenerate only the named fixtures
    tmp = tempfile.mkdtemp(prefix="fg_shims_")
    _install_shims(tmp)
    fg = _import_reference()

    def save(name, d):
        if only and name not in only:
            return
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **d())
        print("%-28s %8.1f kB" % (name, os.path.getsize(path) / 1024))

    # formation_hd_env rollouts, default spread (few contacts) and crowded (many)
    save("hd_n3", lambda: rollout_hd(fg, 3, 4, 25, seed=1, act_seed=11))
    save("hd_n9", lambda: rollout_hd(fg, 9, 4, 25, seed=2, act_seed=12))
    save("hd_n27", lambda: rollout_hd(fg, 27, 3, 25, seed=3, act_seed=13))
    save("hd_n81", lambda: rollout_hd(fg, 81, 2, 25, seed=4, act_seed=14))
    save("hd_n9_crowd", lambda: rollout_hd(fg, 9, 4, 25, seed=5, act_seed=15, crowd=0.15))
    save("hd_n27_crowd", lambda: rollout_hd(fg, 27, 3, 25, seed=6, act_seed=16, crowd=0.25))
    save("hd_n81_crowd", lambda: rollout_hd(fg, 81, 2, 12, seed=7, act_seed=17, crowd=0.3))
    save("hd_n243", lambda: rollout_hd(fg, 243, 1, 3, seed=8, act_seed=18, obs_at=[3]))
    # an even agent count and a non-power-of-3 one (generic-N kernel path)
    save("hd_n4", lambda: rollout_hd(fg, 4, 3, 10, seed=21, act_seed=31))
    save("hd_n10", lambda: rollout_hd(fg, 10, 2, 10, seed=22, act_seed=32, crowd=0.3))
    # more run-time agent counts (lane groups of 8, 16, 64 lanes and a whole-workgroup env), sparse and crowded
    save("hd_n5", lambda: rollout_hd(fg, 5, 3, 12, seed=27, act_seed=37))
    save("hd_n6_crowd", lambda: rollout_hd(fg, 6, 3, 12, seed=28, act_seed=38, crowd=0.1))
    save("hd_n16_crowd", lambda: rollout_hd(fg, 16, 2, 12, seed=29, act_seed=39, crowd=0.2))
    save("hd_n50", lambda: rollout_hd(fg, 50, 2, 8, seed=30, act_seed=40, obs_at=[1, 8]))
    save("hd_n100_crowd", lambda: rollout_hd(fg, 100, 1, 6, seed=31, act_seed=41, crowd=0.35, obs_at=[6]))
    # done flip of formation_hd_env at world_length = 100
    save("hd_n9_options", lambda: rollout_hd(fg, 9, 4, 30, seed=23, act_seed=33, options=dict(max_speed=0.6, accel=3.0, walls=True)))
    save("hd_n27_walls", lambda: rollout_hd(fg, 27, 2, 20, seed=24, act_seed=34, options=dict(walls=True)))
    # non-default World constants (dt, damping, contact force / margin, agent mass and size, episode length)
    save("hd_n9_constants", lambda: rollout_hd(fg, 9, 3, 12, seed=25, act_seed=35, crowd=0.3, obs_at=[1, 12], options=dict(
        world=dict(dt=0.05, damping=0.4, contact_force=60.0, contact_margin=4e-3, mass=2.5, size=0.08, world_length=7))))
    save("hd_n27_constants", lambda: rollout_hd(fg, 27, 2, 8, seed=26, act_seed=36, crowd=0.4, obs_at=[8], options=dict(
        world=dict(dt=0.2, damping=0.1, contact_force=150.0, contact_margin=2e-3, mass=0.5, size=0.05, world_length=5))))
    # per-agent mass / size / accel / max_speed (force_ratio m_b / m_a of core.py:314-317, per-pair contact distances)
    save("hd_n9_masses", lambda: rollout_hd(fg, 9, 3, 16, seed=93, act_seed=94, crowd=0.25, obs_at=[1, 16],
                                            options=dict(hetero=hetero_options(9, 95))))
    save("hd_n27_masses", lambda: rollout_hd(fg, 27, 2, 10, seed=96, act_seed=97, crowd=0.45, obs_at=[10],
                                             options=dict(hetero=hetero_options(27, 98), walls=True)))
    # agents that do not collide, ghosts and a soft wall (core.py:292-293, 326-327), through env.step
    save("hd_n9_flags", lambda: rollout_hd(fg, 9, 2, 14, seed=103, act_seed=104, crowd=0.22, obs_at=[1, 14],
                                           options=dict(walls=True, soft_walls=True, hetero=hetero_options(9, 105),
                                                        flags=dict(collide=np.arange(9) % 4 != 2, ghost=np.arange(9) % 3 == 1))))
    # an immovable agent, driven through the World API (env.step asserts on it)
    save("hd_n6_immovable", lambda: immovable_fixture(fg, 6, 12, seed=107, act_seed=108, crowd=0.2))
    # scripted agents (Agent.action_callback), driven through the World API
    save("hd_n6_scripted", lambda: scripted_fixture(fg, 6, 12, seed=109, act_seed=110, crowd=0.25))
    # non-silent agents: World.step + Scenario.observation with state.c in the communication block
    save("hd_n5_comm", lambda: comm_fixture(fg, 5, 10, seed=99, act_seed=100, crowd=0.3))
    save("hd_n3_done", lambda: rollout_hd(fg, 3, 1, 102, seed=9, act_seed=19, obs_at=[100]))
    # config 1: basic_formation_env, N=3, incl. the done flip at step 50
    save("basic_n3", lambda: rollout_basic(fg, 3, 52, seed=1, act_seed=20))
    save("reset", lambda: reset_fixture(fg, [(1, 3), (7, 9), (1001, 9), (3, 27), (4, 81)]))
    save("shapes", lambda: shape_fixture(fg))
    save("policy_n3", lambda: policy_fixture(fg, 3, 30, seed=41))
    save("policy_n9", lambda: policy_fixture(fg, 9, 30, seed=42))
    save("policy_n27", lambda: policy_fixture(fg, 27, 12, seed=43))
    save("policy_n81", lambda: policy_fixture(fg, 81, 4, seed=44))
    # other hierarchies: test.py -n 2 --num-layer 3, -n 4 --num-layer 2, -n 5 --num-layer 1
    save("policy_n8_per2", lambda: policy_fixture(fg, 8, 10, seed=45, per=2))
    save("policy_n16_per4", lambda: policy_fixture(fg, 16, 8, seed=46, per=4))
    save("policy_n5_per5", lambda: policy_fixture(fg, 5, 10, seed=47, per=5))
    save("benchmark_n9", lambda: benchmark_fixture(fg, 9, 10, seed=91, act_seed=92, crowd=0.12))
    save("hausdorff_kat", lambda: hausdorff_kat())
    # a reference-style plugin file of this repo (not of the reference), executed by the reference's env shell
    plugin = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plugins", "ring_patrol_env.py")
    save("ring_patrol_n5", lambda: plugin_fixture(fg, plugin, 5, 22, seed=61, act_seed=62))
    # the vec-env loop of the trainers (DummyVecEnv.step_wait restated), across an episode end
    save("vec_env_n3", lambda: vec_env_fixture(fg, 3, 3, 104, seed=81, act_seed=82))
    # ... and one whose landmarks collide (movable rocks, an immovable pillar) and whose reward callback writes state
    rocks = os.path.join(os.path.dirname(plugin), "drifting_rocks_env.py")
    save("drifting_rocks_n4", lambda: plugin_fixture(fg, rocks, 4, 15, seed=71, act_seed=72))
    # non-default action modes of _set_action (environment.py:187-216)
    save("act_onehot5_n3", lambda: rollout_action_mode(fg, "onehot5", 3, 8, seed=71, act_seed=81))
    save("act_index_n9", lambda: rollout_action_mode(fg, "index", 9, 8, seed=72, act_seed=82))
    save("act_argmax_n3", lambda: rollout_action_mode(fg, "argmax", 3, 8, seed=73, act_seed=83))
    # remaining scenarios ("next" row f3)
    save("partial_n5", lambda: rollout_scn(fg, "formation_hd_partial_env", 5, 3, 27, seed=51, act_seed=61))
    save("partial_n9_crowd", lambda: rollout_scn(fg, "formation_hd_partial_env", 9, 3, 12, seed=52, act_seed=62, crowd=0.15))
    save("partial_n3", lambda: rollout_scn(fg, "formation_hd_partial_env", 3, 2, 6, seed=53, act_seed=63))
    save("range_n4", lambda: rollout_scn(fg, "formation_hd_partial_range_env", 4, 3, 27, seed=54, act_seed=64))
    save("range_n7_crowd", lambda: rollout_scn(fg, "formation_hd_partial_range_env", 7, 3, 12, seed=55, act_seed=65, crowd=0.2))
    save("obst_n4", lambda: rollout_scn(fg, "formation_hd_obs_env", 4, 3, 52, seed=56, act_seed=66))
    save("obst_n8", lambda: rollout_scn(fg, "formation_hd_obs_env", 8, 2, 40, seed=57, act_seed=67))
    # agents of different mass / size / max_speed among the falling obstacles (force ratio m_b / m_a, contact distance size_a + size_b)
    save("obst_n5_masses", lambda: rollout_scn(fg, "formation_hd_obs_env", 5, 3, 30, seed=58, act_seed=68, crowd=0.35,
                                               hetero=dict(mass=[0.6, 1.0, 2.5, 1.4, 0.8], size=[0.06, 0.1, 0.14, 0.08, 0.12],
                                                           max_speed=[np.nan, 0.5, np.nan, np.nan, 0.8])))
    save("partial_n6_masses", lambda: rollout_scn(fg, "formation_hd_partial_env", 6, 2, 14, seed=59, act_seed=69, crowd=0.12,
                                                  hetero=dict(mass=[0.5, 1.0, 2.0, 3.0, 1.5, 0.7], size=[0.03, 0.05, 0.07, 0.04, 0.06, 0.05],
                                                              max_speed=[np.nan] * 6)))
    # agents that do not collide and ghosts among walls (a soft one too) in the landmark scenarios, through env.step
    save("obst_n5_flags", lambda: rollout_scn(fg, "formation_hd_obs_env", 5, 2, 30, seed=111, act_seed=112, crowd=0.3, walls=True,
                                              hetero=dict(mass=[0.7, 1.0, 2.0, 1.3, 0.9], size=[0.07, 0.1, 0.13, 0.09, 0.11],
                                                          max_speed=[np.nan, 0.6, np.nan, np.nan, np.nan]),
                                              flags=dict(collide=[True, True, False, True, True], ghost=[False, True, False, False, True])))
    save("basic_n4_flags", lambda: rollout_scn(fg, "basic_formation_env", 4, 2, 16, seed=113, act_seed=114, crowd=0.12, walls=True,
                                               hetero=dict(mass=[1.0, 0.6, 1.8, 1.2], size=[0.15, 0.1, 0.2, 0.12],
                                                           max_speed=[np.nan] * 4),
                                               flags=dict(collide=[True, False, True, True], ghost=[True, False, False, False])))
    # ... and with an immovable agent, driven through the World API
    save("obst_n5_immovable", lambda: scn_immovable_fixture(
        fg, "formation_hd_obs_env", 5, 24, seed=115, act_seed=116, crowd=0.3,
        hetero=dict(mass=[0.7, 1.0, 2.0, 1.3, 0.9], size=[0.07, 0.1, 0.13, 0.09, 0.11], max_speed=[np.nan] * 5),
        flags=dict(movable=[True, False, True, True, True], collide=[True, True, True, False, True],
                   ghost=[False, False, True, False, False])))
    save("partial_n6_immovable", lambda: scn_immovable_fixture(
        fg, "formation_hd_partial_env", 6, 14, seed=117, act_seed=118, crowd=0.12,
        hetero=dict(mass=[0.5, 1.0, 2.0, 3.0, 1.5, 0.7], size=[0.03, 0.05, 0.07, 0.04, 0.06, 0.05], max_speed=[np.nan] * 6),
        flags=dict(movable=[True, True, False, True, True, False], collide=[True, True, True, True, False, True],
                   ghost=[False, True, False, False, False, False])))


if __name__ == "__main__":
    main()
