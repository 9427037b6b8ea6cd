"""CPU oracle for the formation_gym hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (NumPy, float64 by default) of the reference's
algorithm for `MultiAgentEnv.step` -> `World.step` -> `Scenario.observation /
reward`.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
leg may import it; the product path (gym-formation_amd/) never does and fails
loudly when the HIP library is missing.

Parity pin: every function below is checked in tests/test_oracle_golden.py
against fixtures captured from the REAL reference (tests/golden/*.npz, made by
tests/golden/make_golden.py in the build container) to <= 1e-12, plus scipy's
published `directed_hausdorff` docstring example.  The reference itself has no
tests or golden vectors for this path.

Third-party arithmetic on the path: scipy.spatial.distance.directed_hausdorff
(unpinned by the reference's setup.py; scipy 1.15.3 in this image).  Its
published semantics - exact max-min Euclidean distance - are restated in
`directed_hausdorff_bruteforce`.

Reference citations are `path:line` under /root/reference/formation_gym/.

Two restatements live here:
  * `PortEnv`   - one env, object-per-agent, Python loops over pairs, reward
                  evaluated 2N times with a Hausdorff max-min each; the same
                  algorithmic structure as the reference.  This is the
                  `cpu_baseline` ("port") that bench.py times.
  * `step_hd` / `step_basic` - batched [B, N] vectorised NumPy; the checker the
                  GPU parity tests compare against.
"""
import math

import numpy as np

# --------------------------------------------------------------------------
# constants in force at the BASELINE configs
# --------------------------------------------------------------------------


class HdParams(object):
    """formation_hd_env constants.  core.py:119-139, formation_hd_env.py:13-33,
    environment.py:218-221."""
    dt = 0.1                 # core.py:125
    damping = 0.25           # core.py:127
    contact_force = 1e2      # core.py:129
    contact_margin = 1e-3    # core.py:130
    sensitivity = 5.0        # environment.py:218 (accel is None)
    mass = 1.0               # core.py:69,73-75
    agent_size = 0.03        # formation_hd_env.py:26
    world_length = 100       # formation_hd_env.py:13,16

    @property
    def dist_min(self):      # core.py:307  size_a + size_b
        return self.agent_size + self.agent_size

    @property
    def collide_thresh(self):  # formation_hd_env.py:121  (size_a + size_b)/2
        return (self.agent_size + self.agent_size) / 2


class BasicParams(HdParams):
    """basic_formation_env constants.  basic_formation_env.py:7-27,
    core.py:52,113."""
    agent_size = 0.1         # basic_formation_env.py:18
    landmark_size = 0.05     # core.py:52 default
    world_length = 50        # core.py:113 default
    num_landmarks = 3        # basic_formation_env.py:7

    @property
    def collide_thresh(self):  # basic_formation_env.py:91  size_a + size_b
        return self.agent_size + self.agent_size


# --------------------------------------------------------------------------
# shared scalar pieces
# --------------------------------------------------------------------------

def softplus_penetration(dist, dist_min, k):
    """core.py:309-310: k * logaddexp(0, -(dist - dist_min)/k), stable form."""
    x = -(dist - dist_min) / k
    return k * (np.maximum(x, 0.0) + np.log1p(np.exp(-np.abs(x))))


def directed_hausdorff_bruteforce(u, v):
    """max_i min_j ||u_i - v_j|| with witness indices (i*, argmin_j D[i*]).
    Restates scipy.spatial.distance.directed_hausdorff's published semantics
    (call sites formation_hd_env.py:66); ties resolve to the first index."""
    u = np.asarray(u, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    D = np.sqrt(((u[:, None, :] - v[None, :, :]) ** 2).sum(-1))
    rowmin = D.min(1)
    i = int(np.argmax(rowmin))
    j = int(np.argmin(D[i]))
    return float(rowmin[i]), i, j


def reset_draws(seed, N, num_landmarks=None):
    """The reference's reset RNG stream (environment.py:106-110 seeds the global
    legacy MT19937; formation_hd_env.py:77-95 draws N agent positions, N landmark
    positions, one ideal velocity, 2 doubles each, in that order).
    Returns (pos[N,2], raw_landmarks[L,2], ideal_vel[2])."""
    L = N if num_landmarks is None else num_landmarks
    rs = np.random.RandomState(seed)
    pos = rs.uniform(-1, +1, (N, 2))
    raw = rs.uniform(-1, +1, (L, 2))
    ivel = rs.uniform(-1, +1, 2)
    return pos, raw, ivel


def reset_hd(seeds, N):
    """Batched formation_hd_env reset (formation_hd_env.py:77-95): one legacy
    RandomState per env.  Returns dict of float64 arrays."""
    B = len(seeds)
    pos = np.zeros((B, N, 2)); shape = np.zeros((B, N, 2)); ivel = np.zeros((B, 2))
    for b, s in enumerate(seeds):
        p, raw, iv = reset_draws(int(s), N)
        pos[b] = p
        shape[b] = raw - raw.mean(0)           # formation_hd_env.py:93
        ivel[b] = iv
    return dict(pos=pos, vel=np.zeros((B, N, 2)), ideal_shape=shape, ideal_vel=ivel,
                step=np.zeros(B, dtype=np.int32))


def reset_basic(seed, N, L=3):
    """basic_formation_env.py:54-65: agents then landmarks, no ideal velocity."""
    rs = np.random.RandomState(seed)
    pos = rs.uniform(-1, +1, (N, 2))
    lm = rs.uniform(-1, +1, (L, 2))
    return dict(pos=pos[None], vel=np.zeros((1, N, 2)), landmarks=lm[None],
                step=np.zeros(1, dtype=np.int32))


# --------------------------------------------------------------------------
# batched vectorised oracle
# --------------------------------------------------------------------------

GOLDEN_WALLS = [("V", -0.9, (-1.0, 1.0), 0.1), ("V", 0.9, (-0.6, 0.6), 0.1), ("H", 0.8, (-0.5, 0.5), 0.2)]
GOLDEN_SOFT_WALLS = [("H", -0.05, (-0.4, 0.4), 0.1)]    # the soft wall of fixture hd_n9_flags (tests/golden/make_golden.py SOFT_WALLS)


def wall_force(pos, size, wall, P, dtype=np.float64, ghost=None):
    """core.py:325-362 get_wall_collision_force for every entity [B,N,2] against one wall
    (orient, axis_pos, endpoints, width[, hard]); a ghost entity passes through a soft wall (:326-327)."""
    orient, axis_pos, ep, width = wall[:4]
    hard = wall[4] if len(wall) > 4 else True
    prll, perp = (0, 1) if orient == "H" else (1, 0)
    x = pos[..., prll]
    beyond = (x < ep[0] - size) | (x > ep[1] + size)
    partial = ~beyond & ((x < ep[0]) | (x > ep[1]))
    past = np.where(x < ep[0], x - ep[0], x - ep[1])
    with np.errstate(invalid="ignore"):
        theta = np.where(partial, np.arcsin(np.clip(past / size, -1, 1)), 0.0)
    dist_min = np.where(partial, np.cos(theta) * size + 0.5 * width, size + 0.5 * width)
    delta = pos[..., perp] - axis_pos
    dist = np.abs(delta)
    pen = softplus_penetration(dist, dist_min, dtype(P.contact_margin))
    with np.errstate(invalid="ignore", divide="ignore"):
        mag = dtype(P.contact_force) * delta / dist * pen
    f = np.zeros_like(pos)
    f[..., perp] = np.cos(theta) * mag
    f[..., prll] = np.sin(theta) * np.abs(mag)
    if ghost is not None and not hard:
        beyond = beyond | np.asarray(ghost, dtype=bool)[None, :]
    return np.where(beyond[..., None], dtype(0), f)


def physics_step(pos, vel, act, P, dtype=np.float64, max_speed=None, accel=None, walls=None, mass=None, size=None,
                 movable=None, collide=None, ghost=None, scripted=None):
    """World.step for agent-only colliders (core.py:206-322 with the early-outs
    of :292-297 applied: landmarks have collide=False, so only agent-agent
    pairs survive).  pos, vel, act: [B,N,2].  Returns new (pos, vel).

    environment.py:216-221  u = sensitivity * action (no clipping)
    core.py:235-236         F_i = mass * u_i   (accel None, no noise)
    core.py:304-318         pair force on PRE-step positions, ratio m_b/m_a
    core.py:268-277         v = v*(1-damping) + F/m*dt ; p += v*dt
    max_speed / accel: a scalar for every agent or one value per agent [N] (NaN = None for that agent);
    mass / size: per-agent arrays [N] (core.py:68-75: Entity.initial_mass, Entity.size), default P.mass / P.agent_size.
    movable / collide / ghost: per-agent booleans [N] (core.py:54-58): a pair needs both to collide and one to move
    (:292-295); against an immovable partner the force is not scaled by the mass ratio (:319-321); an immovable agent
    takes no action force (:231) and is not integrated (:266-267); a ghost passes through soft walls (:326-327).
    scripted: per-agent booleans [N]: `act` of such an agent is a scripted agent's `action.u` (core.py:210-211), used as it
    is - the sensitivity of environment.py:216-221 scales policy agents' actions only.
    """
    pos = np.asarray(pos, dtype=dtype); vel = np.asarray(vel, dtype=dtype)
    act = np.asarray(act, dtype=dtype)
    B, N, _ = pos.shape
    m = np.full(N, P.mass, dtype=dtype) if mass is None else np.asarray(mass, dtype=dtype)
    sz = np.full(N, P.agent_size, dtype=dtype) if size is None else np.asarray(size, dtype=dtype)
    acc = np.full(N, np.nan) if accel is None else np.broadcast_to(np.asarray(accel, dtype=np.float64), (N,))
    has_acc = ~np.isnan(acc)
    sens = np.where(has_acc, acc, P.sensitivity).astype(dtype)           # environment.py:218-220
    if scripted is not None:
        sens = np.where(np.asarray(scripted, dtype=bool), dtype(1), sens)   # core.py:210-211, 235-236
    gain = np.where(has_acc, m * np.where(has_acc, acc, 1.0), m).astype(dtype)   # core.py:236
    F = gain[None, :, None] * (sens[None, :, None] * act)
    delta = pos[:, :, None, :] - pos[:, None, :, :]            # [B,i,j,2] = p_i - p_j
    dist = np.sqrt((delta ** 2).sum(-1))                       # [B,i,j]
    dist_min = (sz[:, None] + sz[None, :]).astype(dtype)       # core.py:307 size_a + size_b
    pen = softplus_penetration(dist, dist_min[None], dtype(P.contact_margin))
    with np.errstate(invalid="ignore", divide="ignore"):
        f = dtype(P.contact_force) * delta / dist[..., None] * pen[..., None]
    mov = np.ones(N, dtype=bool) if movable is None else np.asarray(movable, dtype=bool)
    col = np.ones(N, dtype=bool) if collide is None else np.asarray(collide, dtype=bool)
    both = mov[:, None] & mov[None, :]
    ratio = np.where(both, m[None, :] / m[:, None], 1.0).astype(dtype)   # core.py:314-321: agent i receives (m_j / m_i) f, or f
    f = ratio[None, :, :, None] * f
    pair = col[:, None] & col[None, :] & (mov[:, None] | mov[None, :]) & ~np.eye(N, dtype=bool)   # :292-297
    f = np.where(pair[None, :, :, None], f, dtype(0))
    F = np.where(mov[None, :, None], F, dtype(0)) + f.sum(2)   # :231 an immovable agent takes no action force
    for w in (walls or []):                                    # core.py:255-261
        F = F + wall_force(pos, sz[None, :], w, P, dtype, ghost=ghost)
    new_vel = vel * dtype(1 - P.damping) + (F / m[None, :, None]) * dtype(P.dt)
    if max_speed is not None:                                  # core.py:271-276
        ms = np.broadcast_to(np.asarray(max_speed, dtype=np.float64), (N,))[None, :, None]
        speed = np.sqrt((new_vel ** 2).sum(-1, keepdims=True))
        with np.errstate(invalid="ignore", divide="ignore"):
            new_vel = np.where(speed > ms, new_vel / speed * ms.astype(dtype), new_vel)     # NaN (None) compares False
    new_pos = pos + new_vel * dtype(P.dt)
    keep = ~mov[None, :, None]                                 # :266-267 not movable: state untouched
    return np.where(keep, pos, new_pos), np.where(keep, vel, new_vel)


def update_comm(action_c, silent=None, dtype=np.float64):
    """World.update_agent_state without noise (core.py:279-286): state.c = action.c, zeros for a silent agent.
    action_c [B,N,dim_c]; silent: bool [N] or None (nobody silent)."""
    c = np.array(action_c, dtype=dtype)
    if silent is not None:
        c[:, np.asarray(silent, dtype=bool)] = 0
    return c


ACT_ONEHOT5, ACT_INDEX, ACT_ARGMAX = 1, 2, 3


def decode_actions(action, mode):
    """MultiAgentEnv._set_action for the non-default action modes, before the sensitivity
    scaling (environment.py:187-216).  Returns raw u [..., 2] (float64).
      ACT_ONEHOT5  discrete_action_space  (:207-210)  action [..., 5]: u = (a1 - a2, a3 - a4)
      ACT_INDEX    discrete_action_input  (:194-205)  action [...] int: 1 -x, 2 +x, 3 -y, 4 +y
      ACT_ARGMAX   force_discrete_action  (:212-216)  action [..., 2]: one-hot of np.argmax"""
    a = np.asarray(action)
    if mode == ACT_ONEHOT5:
        a = a.astype(np.float64)
        return np.stack([a[..., 1] - a[..., 2], a[..., 3] - a[..., 4]], -1)
    if mode == ACT_INDEX:
        u = np.zeros(a.shape + (2,))
        u[..., 0] = np.where(a == 1, -1.0, np.where(a == 2, 1.0, 0.0))
        u[..., 1] = np.where(a == 3, -1.0, np.where(a == 4, 1.0, 0.0))
        return u
    if mode == ACT_ARGMAX:
        p = np.argmax(a[..., 0:2], axis=-1)
        return np.stack([(p == 0).astype(np.float64), (p == 1).astype(np.float64)], -1)
    raise ValueError("unknown action mode %r" % (mode,))


def observation_hd(pos, vel, ideal_shape, ideal_vel, dtype=np.float64, comm=None):
    """formation_hd_env.py:52-59 for every agent: [v_i | p_j - p_i (j != i, index
    order) | c_j (j != i; zeros for silent agents) 2(N-1) | ideal_shape.flatten() | ideal_vel] -> [B,N,6N].
    comm [B,N,2] = AgentState.c of every agent (dim_c = 2), None = all silent."""
    pos = np.asarray(pos, dtype=dtype); vel = np.asarray(vel, dtype=dtype)
    B, N, _ = pos.shape
    obs = np.zeros((B, N, 6 * N), dtype=dtype)
    obs[:, :, 0:2] = vel
    rel = pos[:, None, :, :] - pos[:, :, None, :]              # [B,i,j,2] = p_j - p_i
    keep = ~np.eye(N, dtype=bool)
    obs[:, :, 2:2 * N] = rel[:, keep].reshape(B, N, 2 * (N - 1))
    if comm is not None:                                       # :48-51 comm = append(other.state.c) for other != agent
        c = np.broadcast_to(np.asarray(comm, dtype=dtype)[:, None, :, :], (B, N, N, 2))
        obs[:, :, 2 * N:4 * N - 2] = c[:, keep].reshape(B, N, 2 * (N - 1))
    obs[:, :, 4 * N - 2:6 * N - 2] = np.asarray(ideal_shape, dtype=dtype).reshape(B, 1, 2 * N)
    obs[:, :, 6 * N - 2:] = np.asarray(ideal_vel, dtype=dtype)[:, None, :]
    return obs


def reward_hd(pos, vel, ideal_shape, ideal_vel, P, dtype=np.float64, size=None, collide=None):
    """formation_hd_env.py:61-75 for every agent + the integer by-products.
    Returns dict(indiv[B,N], shared[B], hd[B,2], hd_idx[B,4], near_lm[B,N],
    near_ag[B,N], cnt[B,N], gap_lm, gap_ag, cnt_margin)."""
    pos = np.asarray(pos, dtype=dtype); vel = np.asarray(vel, dtype=dtype)
    S = np.asarray(ideal_shape, dtype=dtype); iv = np.asarray(ideal_vel, dtype=dtype)
    B, N, _ = pos.shape
    pt = pos - pos.mean(1, keepdims=True)                      # :65
    D = np.sqrt(((pt[:, :, None, :] - S[:, None, :, :]) ** 2).sum(-1))   # [B,i(agent),j(shape)]
    rowmin = D.min(2); colmin = D.min(1)
    h1 = rowmin.max(1); h2 = colmin.max(1)
    H = np.maximum(h1, h2)                                     # :66
    velterm = np.sqrt(((iv - vel.mean(1)) ** 2).sum(-1))       # :68-69
    PD = np.sqrt(((pos[:, :, None, :] - pos[:, None, :, :]) ** 2).sum(-1))
    if size is None:
        thr = dtype(P.collide_thresh)
    else:                                                      # :119-121 per pair: (size_a + size_b) / 2, scaled like P's
        sz = np.asarray(size, dtype=dtype)
        thr = (dtype(P.collide_thresh / P.dist_min) * (sz[:, None] + sz[None, :]))[None]
    close = PD < thr                                           # :121 strict <
    close[:, np.arange(N), np.arange(N)] = False               # :73 agent != a
    cnt = close.sum(2)
    if collide is not None:                                    # :71 `if agent.collide:` - the penalties of a non-collider are not counted
        cnt = np.where(np.asarray(collide, dtype=bool)[None, :], cnt, 0)
    indiv = -H[:, None] - velterm[:, None] - cnt               # :66,:69,:74
    shared = indiv.sum(1)                                      # environment.py:136
    near_lm = D.argmin(2); near_ag = D.argmin(1)
    i1 = rowmin.argmax(1); j1 = near_lm[np.arange(B), i1]
    i2 = colmin.argmax(1); j2 = near_ag[np.arange(B), i2]      # (shape idx, agent idx)
    Ds = np.sort(D, axis=2); Ds0 = np.sort(D, axis=1)
    off = np.abs(PD - thr) + np.eye(N)[None]
    rs = np.sort(rowmin, 1); cs = np.sort(colmin, 1)
    ar = np.arange(B)
    # margin by which the witness indices are determined (max side and its inner argmin)
    hd_gap = np.stack([np.minimum(rs[:, -1] - rs[:, -2], (Ds[:, :, 1] - Ds[:, :, 0])[ar, i1]),
                       np.minimum(cs[:, -1] - cs[:, -2], (Ds0[:, 1, :] - Ds0[:, 0, :])[ar, i2])], 1)
    return dict(indiv=indiv, shared=shared, hd=np.stack([h1, h2], 1), hd_gap=hd_gap,
                hd_idx=np.stack([i1, j1, i2, j2], 1).astype(np.int32),
                near_lm=near_lm.astype(np.int32), near_ag=near_ag.astype(np.int32),
                cnt=cnt.astype(np.int32), gap_lm=Ds[:, :, 1] - Ds[:, :, 0],
                gap_ag=Ds0[:, 1, :] - Ds0[:, 0, :], cnt_margin=off.reshape(B, -1).min(1),
                velterm=velterm, H=H)


def step_hd(state, act, P=None, dtype=np.float64, **world_options):
    """One MultiAgentEnv.step of formation_hd_env for B envs (environment.py:
    113-142).  `state` = dict(pos, vel, ideal_shape, ideal_vel, step); returns
    (new_state, out) with out = dict(obs, reward[B,N,1], done[B,N], indiv, ...)."""
    P = P or HdParams()
    comm = world_options.pop("comm", None)                     # AgentState.c [B,N,2] of non-silent agents
    pos, vel = physics_step(state["pos"], state["vel"], act, P, dtype, **world_options)
    step = np.asarray(state["step"]) + 1                       # environment.py:114
    out = reward_hd(pos, vel, state["ideal_shape"], state["ideal_vel"], P, dtype, size=world_options.get("size"),
                    collide=world_options.get("collide"))
    out["obs"] = observation_hd(pos, vel, state["ideal_shape"], state["ideal_vel"], dtype, comm=comm)
    N = pos.shape[1]
    out["reward"] = np.repeat(out["shared"][:, None], N, 1)[..., None]    # :136-138
    out["done"] = np.repeat((step >= P.world_length)[:, None], N, 1)      # :172-178
    new_state = dict(state, pos=pos, vel=vel, step=step.astype(np.int32))
    return new_state, out


def benchmark_data_hd(pos, landmarks, indiv, P):
    """Scenario.benchmark_data (formation_hd_env.py:97-117) for every agent of every env.
    pos [B,N,2], landmarks [B,L,2] (as they stand when the env calls it: re-centred on the agents by
    `observation`, :40-44), indiv [B,N] = Scenario.reward.  collisions counts every agent within the reward's
    contact distance INCLUDING the agent itself (:101-104 has no `a is agent` guard)."""
    d = np.sqrt(((pos[:, :, None, :] - pos[:, None, :, :]) ** 2).sum(-1))
    collisions = (d < P.collide_thresh).sum(2)
    dl = np.sqrt(((pos[:, :, None, :] - landmarks[:, None, :, :]) ** 2).sum(-1)).min(1)      # per landmark: nearest agent
    B, N = pos.shape[:2]
    return dict(reward=indiv, collisions=collisions,
                min_dists=np.repeat(dl.sum(1)[:, None], N, 1),
                occupied_landmarks=np.repeat((dl < 0.1).sum(1)[:, None], N, 1))


def observation_basic(pos, vel, landmarks, dtype=np.float64):
    """basic_formation_env.py:29-41: [v_i | p_i | l_k - p_i | p_j - p_i (j != i) |
    zeros 2(N-1)] -> [B,N,4+2L+4(N-1)]."""
    pos = np.asarray(pos, dtype=dtype); vel = np.asarray(vel, dtype=dtype)
    lm = np.asarray(landmarks, dtype=dtype)
    B, N, _ = pos.shape; L = lm.shape[1]
    dim = 4 + 2 * L + 4 * (N - 1)
    obs = np.zeros((B, N, dim), dtype=dtype)
    obs[:, :, 0:2] = vel
    obs[:, :, 2:4] = pos
    obs[:, :, 4:4 + 2 * L] = (lm[:, None, :, :] - pos[:, :, None, :]).reshape(B, N, 2 * L)
    rel = pos[:, None, :, :] - pos[:, :, None, :]
    keep = ~np.eye(N, dtype=bool)
    obs[:, :, 4 + 2 * L:4 + 2 * L + 2 * (N - 1)] = rel[:, keep].reshape(B, N, 2 * (N - 1))
    return obs


def reward_basic(pos, landmarks, P, dtype=np.float64, size=None, collide=None):
    """basic_formation_env.py:43-52: -sum_l min_a ||p_a - l|| - #{a (self
    included): ||p_a - p_i|| < size_a + size_i}; the count only `if agent.collide:` (:48).  size / collide: per agent [N]."""
    pos = np.asarray(pos, dtype=dtype); lm = np.asarray(landmarks, dtype=dtype)
    B, N, _ = pos.shape
    D = np.sqrt(((pos[:, :, None, :] - lm[:, None, :, :]) ** 2).sum(-1))   # [B,a,l]
    cover = D.min(1).sum(1)
    PD = np.sqrt(((pos[:, :, None, :] - pos[:, None, :, :]) ** 2).sum(-1))
    thr = dtype(P.collide_thresh)
    if size is not None:
        sz = np.asarray(size, dtype=dtype)
        thr = (sz[:, None] + sz[None, :])[None]
    cnt = (PD < thr).sum(2)                                    # includes self
    if collide is not None:
        cnt = np.where(np.asarray(collide, dtype=bool)[None, :], cnt, 0)
    indiv = -cover[:, None] - cnt
    return dict(indiv=indiv, shared=indiv.sum(1), cnt=cnt.astype(np.int32),
                near_ag=D.argmin(1).astype(np.int32))


def step_basic(state, act, P=None, dtype=np.float64, **world_options):
    """One env.step of basic_formation_env; world_options as `physics_step` takes them (per-agent mass / size / flags, walls)."""
    P = P or BasicParams()
    pos, vel = physics_step(state["pos"], state["vel"], act, P, dtype, **world_options)
    step = np.asarray(state["step"]) + 1
    out = reward_basic(pos, state["landmarks"], P, dtype, size=world_options.get("size"), collide=world_options.get("collide"))
    out["obs"] = observation_basic(pos, vel, state["landmarks"], dtype)
    N = pos.shape[1]
    out["reward"] = np.repeat(out["shared"][:, None], N, 1)[..., None]
    out["done"] = np.repeat((step >= P.world_length)[:, None], N, 1)
    return dict(state, pos=pos, vel=vel, step=step.astype(np.int32)), out


def generate_shape(layer):
    """formation_hd_env.py:123-139 default hierarchical shape, returned [3^(layer+1), 2]."""
    table = np.array([
        [[0, -1], [0.5, 0], [0, 1]],
        [[0, 1.6], [-1, 0], [1, 0]],
        [[1.5, 0], [0, 0], [-1.5, 0]],
        [[0, 0.6], [1, 0], [-1, 0]],
    ], dtype=np.float64)
    assert layer < table.shape[0], "Layer shape is not enough!"
    pts = table[0]
    for l in range(1, layer + 1):
        pts = np.concatenate([table[l][i] + 0.45 * pts for i in range(3)], 0)
    return pts


# --------------------------------------------------------------------------
# demo policies (reference __init__.py:19-99), "next" row f2
# --------------------------------------------------------------------------

def ezpolicy(obs):
    """__init__.py:19-47 hand-written formation controller on one observation."""
    obs = np.asarray(obs, dtype=np.float64)
    n = len(obs) / 6
    assert float(n).is_integer(), n
    n = int(n)
    others = obs[2:2 * n]
    ideal = obs[4 * n - 2:6 * n - 2].reshape(-1, 2)
    ideal = ideal - ideal.mean(0)
    ivel = obs[-2:]
    cur = np.append(others, [0.0, 0.0]).reshape(-1, 2)
    cur = cur - cur.mean(0)
    me = cur[-1]
    order = np.argsort(np.sqrt(((me - ideal) ** 2).sum(1)), kind="quicksort")
    act = None
    for idx in order:
        closest = int(np.argmin(np.sqrt(((cur - ideal[idx]) ** 2).sum(1))))
        if closest == n - 1 or idx == order[-1]:
            act = np.clip(0.5 * (ideal[idx] - me), -1, 1)
            break
    done = np.linalg.norm(ideal - cur) < 0.01
    return act + (ivel if done else 0.3 * ivel)


def get_action_bfs(policy, obs, per_layer, strict=True):
    """__init__.py:49-99 breadth-first hierarchical expansion of `policy`.
    strict: keep the reference's own shape test (:55-56), which compares a float log ratio with an integer and
    therefore REJECTS 3^5 = 243, 5^3 = 125 and 6^3 = 216 agents (np.log(243)/np.log(3) = 4.999999999999999);
    strict=False accepts every exact power (the build's kernel does) so that those sizes can be checked too."""
    layers = np.log(len(obs)) / np.log(per_layer)
    if strict:
        assert float(layers).is_integer(), "Observation shape error!"
    else:
        assert per_layer ** int(round(layers)) == len(obs), "Observation shape error!"
    queue = [[np.asarray(o, dtype=np.float64) for o in obs]]
    acts = []
    while queue:
        group = queue.pop(0)
        n_cur = len(group)
        n_sub = n_cur // per_layer
        level = np.log(n_cur) / np.log(per_layer)
        for i in range(per_layer):
            lead = group[i * n_sub]
            rel = np.insert(lead[2:2 * n_cur], 2 * i * n_sub, [0.0, 0.0]).reshape(-1, 2)
            cent = np.array([rel[n_sub * k:n_sub * (k + 1)].mean(0) for k in range(per_layer)])
            cent = np.delete(cent - cent[i], i, 0).ravel()
            ideal = lead[4 * n_cur - 2:6 * n_cur - 2].reshape(-1, 2)
            tgt = np.array([ideal[n_sub * k:n_sub * (k + 1)].mean(0) for k in range(per_layer)]).ravel()
            inp = np.concatenate((lead[:2], cent, np.zeros(2 * (per_layer - 1)), tgt, lead[-2:]))
            sub_vel = policy(inp) * level
            if n_sub == 1:
                acts.append(sub_vel)
                continue
            nxt = []
            for j in range(i * n_sub, (i + 1) * n_sub):
                o = group[j]
                oth = o[2:2 * n_cur][2 * i * n_sub:2 * (i + 1) * n_sub - 2]
                shp = o[4 * n_cur - 2:6 * n_cur - 2][2 * i * n_sub:2 * (i + 1) * n_sub]
                nxt.append(np.concatenate((o[:2], oth, np.zeros(2 * (n_sub - 1)), shp, sub_vel)))
            queue.append(nxt)
    return acts


def ezpolicy_margin(obs):
    """Smallest gap of any comparison `ezpolicy` (__init__.py:35-42) makes on this observation: adjacent
    distances of the argsort (:35), me-vs-closest-other per mark (:37-38), the 0.01 formation test (:42).
    A fp32 evaluation may legitimately decide differently only where this is ~1e-6 or less."""
    obs = np.asarray(obs, dtype=np.float64)
    n = len(obs) // 6
    ideal = obs[4 * n - 2:6 * n - 2].reshape(-1, 2)
    ideal = ideal - ideal.mean(0)
    cur = np.append(obs[2:2 * n], [0.0, 0.0]).reshape(-1, 2)
    cur = cur - cur.mean(0)
    d_me = np.sqrt(((cur[-1] - ideal) ** 2).sum(1))
    gaps = [np.diff(np.sort(d_me)).min()] if n > 1 else []
    for k in range(n):
        d = np.sqrt(((cur - ideal[k]) ** 2).sum(1))
        gaps.append(abs(d[-1] - d[:-1].min()))
    gaps.append(abs(np.linalg.norm(ideal - cur) - 0.01))
    return float(min(gaps))


def bfs_margins(obs, per_layer):
    """Per agent: the smallest `ezpolicy_margin` along the chain of decisions that produce its action in
    `get_action_bfs` (the group problems of every level it belongs to)."""
    margins = []

    def policy(inp):
        margins.append(ezpolicy_margin(inp))
        return ezpolicy(inp)

    get_action_bfs(policy, obs, per_layer, strict=False)
    N = len(obs)
    L = int(round(np.log(N) / np.log(per_layer)))
    out = np.full(N, np.inf)
    # `policy` is called level by level, sub-groups in index order (the queue is breadth-first)
    pos = 0
    for lev in range(L, 0, -1):
        n_sub = per_layer ** (lev - 1)
        for sg in range(N // n_sub):
            out[sg * n_sub:(sg + 1) * n_sub] = np.minimum(out[sg * n_sub:(sg + 1) * n_sub], margins[pos])
            pos += 1
    assert pos == len(margins)
    return out


# --------------------------------------------------------------------------
# faithful per-env port: the `cpu_baseline` ("port")
# --------------------------------------------------------------------------

class _Body(object):
    __slots__ = ("pos", "vel", "c", "u", "size", "mass", "movable", "collide")

    def __init__(self, size, movable, collide):
        self.pos = np.zeros(2); self.vel = np.zeros(2); self.c = np.zeros(2)
        self.u = np.zeros(2)
        self.size = size; self.mass = 1.0; self.movable = movable; self.collide = collide


class PortEnv(object):
    """One formation_hd_env instance with the reference's algorithmic structure:
    a Python object per agent/landmark, a Python loop over all entity pairs
    (core.py:240-262), and the reward callback evaluated twice per agent per
    step (environment.py:128,130), each evaluation doing two directed Hausdorff
    passes (formation_hd_env.py:66).  Used only as the timed CPU baseline and as
    an independent cross-check of the vectorised oracle."""

    def __init__(self, num_agents=3, P=None, use_scipy=True):
        self.P = P or HdParams()
        self.N = num_agents
        self.agents = [_Body(self.P.agent_size, True, True) for _ in range(num_agents)]
        self.landmarks = [_Body(0.01, False, False) for _ in range(num_agents)]
        self.t = 0
        self.ideal_shape = np.zeros((num_agents, 2)); self.ideal_vel = np.zeros(2)
        self._hd = None
        if use_scipy:
            try:
                from scipy.spatial.distance import directed_hausdorff
                self._hd = lambda a, b: directed_hausdorff(a, b)[0]
            except Exception:
                self._hd = None
        if self._hd is None:
            self._hd = lambda a, b: directed_hausdorff_bruteforce(a, b)[0]
        self._rs = np.random.RandomState(1)

    def seed(self, seed=None):                                  # environment.py:106-110
        self._rs = np.random.RandomState(1 if seed is None else seed)

    def reset(self):                                            # environment.py:144-156
        self.t = 0
        for a in self.agents:
            a.pos = self._rs.uniform(-1, +1, 2); a.vel = np.zeros(2); a.c = np.zeros(2)
        raw = []
        for l in self.landmarks:
            p = self._rs.uniform(-1, +1, 2)
            raw.append(p); l.pos = p
        self.ideal_shape = np.array(raw) - np.mean(raw, 0)
        self.ideal_vel = self._rs.uniform(-1, +1, 2)
        return [self._obs(a) for a in self.agents]

    def load(self, pos, vel, ideal_shape, ideal_vel, t=0):
        for i, a in enumerate(self.agents):
            a.pos = np.array(pos[i], dtype=np.float64); a.vel = np.array(vel[i], dtype=np.float64)
        self.ideal_shape = np.array(ideal_shape, dtype=np.float64)
        self.ideal_vel = np.array(ideal_vel, dtype=np.float64)
        self.t = t

    # -- World.step ----------------------------------------------------
    def _pair_force(self, a, b):                                # core.py:289-322
        if (not a.collide) or (not b.collide):
            return None, None
        if (not a.movable) and (not b.movable):
            return None, None
        if a is b:
            return None, None
        d = a.pos - b.pos
        dist = math.sqrt(d[0] * d[0] + d[1] * d[1])
        k = self.P.contact_margin
        pen = np.logaddexp(0, -(dist - (a.size + b.size)) / k) * k
        f = self.P.contact_force * d / dist * pen
        r = b.mass / a.mass
        return r * f, -(1 / r) * f

    def _world_step(self):
        ents = self.agents + self.landmarks
        force = [None] * len(ents)
        for i, a in enumerate(self.agents):                     # core.py:228-237
            force[i] = a.mass * a.u
        for ia in range(len(ents)):                             # core.py:240-262
            for ib in range(ia + 1, len(ents)):
                fa, fb = self._pair_force(ents[ia], ents[ib])
                if fa is not None:
                    force[ia] = fa + (0.0 if force[ia] is None else force[ia])
                if fb is not None:
                    force[ib] = fb + (0.0 if force[ib] is None else force[ib])
        for i, e in enumerate(ents):                            # core.py:264-277
            if not e.movable:
                continue
            e.vel = e.vel * (1 - self.P.damping)
            if force[i] is not None:
                e.vel = e.vel + (force[i] / e.mass) * self.P.dt
            e.pos = e.pos + e.vel * self.P.dt
        for a in self.agents:                                   # core.py:279-282
            a.c = np.zeros(2)

    # -- scenario callbacks ---------------------------------------------
    def _obs(self, me):                                         # formation_hd_env.py:38-59
        cen = np.mean([a.pos for a in self.agents], 0) - np.mean([l.pos for l in self.landmarks], 0)
        for l in self.landmarks:
            l.pos = l.pos + cen
        rel = np.array([]); comm = np.array([])
        for o in self.agents:
            if o is me:
                continue
            comm = np.append(comm, o.c)
            rel = np.append(rel, o.pos - me.pos)
        return np.concatenate((me.vel, rel, comm, self.ideal_shape.flatten(), self.ideal_vel))

    def _reward(self, me):                                      # formation_hd_env.py:61-75
        shp = np.array([a.pos for a in self.agents])
        shp = shp - np.mean(shp, 0)
        r = -max(self._hd(shp, self.ideal_shape), self._hd(self.ideal_shape, shp))
        mv = np.mean([a.vel for a in self.agents], axis=0)
        r -= np.linalg.norm(self.ideal_vel - mv)
        for a in self.agents:
            if a is not me and np.linalg.norm(a.pos - me.pos) < (a.size + me.size) / 2:
                r -= 1
        return r

    def step(self, action_n):                                   # environment.py:113-142
        self.t += 1
        for a, u in zip(self.agents, action_n):
            a.u = np.asarray(u, dtype=np.float64) * self.P.sensitivity
        self._world_step()
        obs_n, rew_n, done_n, info_n = [], [], [], []
        for a in self.agents:
            obs_n.append(self._obs(a))
            rew_n.append([self._reward(a)])
            done_n.append(self.t >= self.P.world_length)
            info_n.append({"individual_reward": self._reward(a)})
        total = np.sum(rew_n)
        rew_n = [[total]] * self.N
        return obs_n, rew_n, done_n, info_n


# --------------------------------------------------------------------------
# remaining scenarios ("next" row f3): formation_hd_partial_env,
# formation_hd_partial_range_env, formation_hd_obs_env
# --------------------------------------------------------------------------

class ScnParams(HdParams):
    """Constants of the three landmark-formation scenarios.
    partial:  envs/formation_hd_partial_env.py:15-36   (agent 0.04, num_obs 3, L 5, length 25)
    range:    envs/formation_hd_partial_range_env.py:15-36 (agent 0.04, obs_range 0.7, L 4, length 25)
    obstacle: envs/formation_hd_obs_env.py:14-42       (agent 0.1, L 4, 3 obstacles of 0.15, length 50)"""

    def __init__(self, kind):
        assert kind in ("partial", "range", "obstacle")
        self.kind = kind
        self.agent_size = 0.1 if kind == "obstacle" else 0.04
        self.world_length = 50 if kind == "obstacle" else 25
        self.num_landmarks = {"partial": 5, "range": 4, "obstacle": 4}[kind]
        self.num_obstacles = 3 if kind == "obstacle" else 0
        self.obstacle_size = 0.15
        self.num_obs = 3
        self.obs_range = 0.7
        self.penalty = 2.0 if kind == "obstacle" else 1.0      # formation_hd_obs_env.py:92-98
        self.obstacle_vel = (0.0, -1.0)                        # :84-89
        self.obstacle_floor = -2.2

    @property
    def collide_thresh(self):          # is_collision: size_a + size_b (no /2 in these files)
        return self.agent_size + self.agent_size


def reset_scn(kind, seed, N):
    """reset_world draw order: N agent positions, L landmark positions (U(-1,1)^2); the
    obstacle scenario then draws each obstacle from U([step_k, 2.0], [step_k+1, 2.5])
    (formation_hd_obs_env.py:101-114)."""
    P = ScnParams(kind)
    rs = np.random.RandomState(seed)
    pos = rs.uniform(-1, +1, (N, 2))
    L, M = P.num_landmarks, P.num_obstacles
    lm = np.zeros((L, 2)); ob = np.zeros((M, 2))
    step = np.linspace(-1.8, 1.8, M + 1) if M else None
    for i in range(L + M):
        if i < L:
            lm[i] = rs.uniform(-1, +1, 2)
        else:
            k = i - L
            ob[k] = rs.uniform([step[k], 2.0], [step[k + 1], 2.5])
    return dict(pos=pos[None], vel=np.zeros((1, N, 2)), landmarks=lm[None], obst_pos=ob[None],
                obst_vel=np.tile(np.array(P.obstacle_vel), (1, M, 1)), step=np.zeros(1, dtype=np.int32))


def physics_entities(pos, vel, force0, size, P, mass=None, max_speed=None, movable=None, collide=None, ghost=None, walls=None):
    """World.step over E colliding entities with per-entity size and mass
    (core.py:240-277; force ratio m_b / m_a :314-317; speed clamp :271-276, NaN = None).  force0 = non-contact force per entity.
    movable / collide / ghost: per-entity booleans [E] (core.py:54-58), as in `physics_step`: a pair needs both to collide and
    one to move (:292-295), the force against an immovable partner is not scaled by the mass ratio (:319-321), an immovable
    entity takes no action force (:231) and keeps its state (:266-267), a ghost passes through soft walls (:326-327).
    walls: as in `physics_step` (core.py:255-261)."""
    pos = np.asarray(pos, dtype=np.float64); vel = np.asarray(vel, dtype=np.float64)
    E = pos.shape[1]
    m = np.full(E, P.mass) if mass is None else np.asarray(mass, dtype=np.float64)
    mov = np.ones(E, dtype=bool) if movable is None else np.asarray(movable, dtype=bool)
    col = np.ones(E, dtype=bool) if collide is None else np.asarray(collide, dtype=bool)
    delta = pos[:, :, None, :] - pos[:, None, :, :]
    dist = np.sqrt((delta ** 2).sum(-1))
    dmin = size[:, None] + size[None, :]
    pen = softplus_penetration(dist, dmin[None], P.contact_margin)
    with np.errstate(invalid="ignore", divide="ignore"):
        f = P.contact_force * delta / dist[..., None] * pen[..., None]
    ratio = np.where(mov[:, None] & mov[None, :], m[None, :] / m[:, None], 1.0)
    f = ratio[None, :, :, None] * f
    pair = col[:, None] & col[None, :] & (mov[:, None] | mov[None, :]) & ~np.eye(E, dtype=bool)
    f = np.where(pair[None, :, :, None], f, 0.0)
    F = np.where(mov[None, :, None], force0, 0.0) + f.sum(2)
    for w in (walls or []):
        F = F + wall_force(pos, size[None, :], w, P, np.float64, ghost=ghost)
    new_vel = vel * (1 - P.damping) + (F / m[None, :, None]) * P.dt
    if max_speed is not None:
        ms = np.asarray(max_speed, dtype=np.float64)[None, :, None]
        speed = np.sqrt((new_vel ** 2).sum(-1, keepdims=True))
        with np.errstate(invalid="ignore", divide="ignore"):
            new_vel = np.where(speed > ms, new_vel / speed * ms, new_vel)
    keep = ~mov[None, :, None]
    return np.where(keep, pos, pos + new_vel * P.dt), np.where(keep, vel, new_vel)


def _hausdorff_centred(pos, lm):
    u = pos - pos.mean(1, keepdims=True)
    v = lm - lm.mean(1, keepdims=True)
    D = np.sqrt(((u[:, :, None, :] - v[:, None, :, :]) ** 2).sum(-1))
    return np.maximum(D.min(2).max(1), D.min(1).max(1))


def observation_scn(kind, pos, vel, lm, obst_pos, P):
    B, N, _ = pos.shape
    rel = pos[:, None, :, :] - pos[:, :, None, :]              # [B,i,j] = p_j - p_i
    keep = ~np.eye(N, dtype=bool)
    lm_abs = np.repeat(lm.reshape(B, 1, -1), N, 1)
    zeros = np.zeros((B, N, 2 * (N - 1)))
    if kind == "partial":            # formation_hd_partial_env.py:38-57
        idx = (np.arange(N)[:, None] + 1 + np.arange(P.num_obs)[None, :]) % N
        nb = rel[:, np.arange(N)[:, None], idx].reshape(B, N, 2 * P.num_obs)
        return np.concatenate((vel, lm_abs, nb, zeros), 2)
    if kind == "range":              # formation_hd_partial_range_env.py:38-52
        oth = np.clip(rel[:, keep].reshape(B, N, 2 * (N - 1)), -P.obs_range, P.obs_range)
        return np.concatenate((vel, lm_abs, oth, zeros), 2)
    ob_rel = (obst_pos[:, None, :, :] - pos[:, :, None, :]).reshape(B, N, -1)   # formation_hd_obs_env.py:44-58
    return np.concatenate((vel, lm_abs, ob_rel, rel[:, keep].reshape(B, N, 2 * (N - 1)), zeros), 2)


def step_scn(kind, state, act, P=None, mass=None, size=None, max_speed=None, movable=None, collide=None, ghost=None, walls=None):
    """One env.step of the three scenarios.  state: pos, vel [B,N,2], landmarks [B,L,2],
    obst_pos, obst_vel [B,M,2], step [B].  mass / size / max_speed: per-agent arrays [N] (core.py:45-109; the obstacles keep
    the scenario's size and unit mass).  movable / collide / ghost: per-agent booleans [N] (core.py:54-58; the obstacles are
    ordinary movable colliders, formation_hd_obs_env.py:36-42); the penalties of an agent that does not collide are not
    counted (`if agent.collide:`, formation_hd_partial_env.py:67, formation_hd_obs_env.py:91).  walls: core.py:255-261."""
    P = P or ScnParams(kind)
    pos = np.asarray(state["pos"], dtype=np.float64); vel = np.asarray(state["vel"], dtype=np.float64)
    act = np.asarray(act, dtype=np.float64)
    B, N, _ = pos.shape
    M = P.num_obstacles
    op = np.asarray(state["obst_pos"], dtype=np.float64).reshape(B, M, 2)
    ov = np.asarray(state["obst_vel"], dtype=np.float64).reshape(B, M, 2)
    asz = np.full(N, P.agent_size) if size is None else np.asarray(size, dtype=np.float64)
    am = np.full(N, P.mass) if mass is None else np.asarray(mass, dtype=np.float64)
    size = np.concatenate((asz, [P.obstacle_size] * M))
    F0 = np.concatenate((am[None, :, None] * P.sensitivity * act, np.zeros((B, M, 2))), 1)
    ems = None if max_speed is None else np.concatenate((np.asarray(max_speed, dtype=np.float64), [np.nan] * M))
    ent = lambda flag, fill: None if flag is None else np.concatenate((np.asarray(flag, dtype=bool), [fill] * M))
    ep, ev = physics_entities(np.concatenate((pos, op), 1), np.concatenate((vel, ov), 1), F0, size, P,
                              mass=np.concatenate((am, [1.0] * M)), max_speed=ems, movable=ent(movable, True),
                              collide=ent(collide, True), ghost=ent(ghost, False), walls=walls)
    pos, vel, op, ov = ep[:, :N], ev[:, :N], ep[:, N:], ev[:, N:]
    step = np.asarray(state["step"]) + 1
    lm = np.asarray(state["landmarks"], dtype=np.float64)
    H = _hausdorff_centred(pos, lm)
    PD = np.sqrt(((pos[:, :, None, :] - pos[:, None, :, :]) ** 2).sum(-1))
    close = PD < (asz[:, None] + asz[None, :])[None]           # is_collision: dist < size_a + size_b
    close[:, np.arange(N), np.arange(N)] = False
    counted = np.ones(N, dtype=bool) if collide is None else np.asarray(collide, dtype=bool)
    indiv = -H[:, None] - P.penalty * np.where(counted[None, :], close.sum(2), 0)
    if M:
        # reward side effect: obstacles keep falling until the floor (formation_hd_obs_env.py:84-89)
        ov = np.where((op[..., 1] > P.obstacle_floor)[..., None], np.array(P.obstacle_vel), 0.0)
        OD = np.sqrt(((pos[:, :, None, :] - op[:, None, :, :]) ** 2).sum(-1))
        indiv = indiv - P.penalty * np.where(counted[None, :], (OD < asz[None, :, None] + P.obstacle_size).sum(2), 0)
    out = dict(indiv=indiv, shared=indiv.sum(1), obs=observation_scn(kind, pos, vel, lm, op, P))
    out["reward"] = np.repeat(out["shared"][:, None], N, 1)[..., None]
    out["done"] = np.repeat((step >= P.world_length)[:, None], N, 1)
    new = dict(state, pos=pos, vel=vel, obst_pos=op, obst_vel=ov, step=step.astype(np.int32))
    return new, out
