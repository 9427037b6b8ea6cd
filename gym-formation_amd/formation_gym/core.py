"""Batched physics core: the MI355X-native counterpart of reference
formation_gym/core.py.

The reference keeps one Python object graph per environment, each agent holding
`np.ndarray(2,)` float64 state (core.py:4-24, :45-109), and `World.step()`
(core.py:206-225) is a pile of Python loops.  Here a `World` holds B independent
environments as structure-of-arrays float32 tensors on the GPU:

    pos_x, pos_y, vel_x, vel_y   [B, N]      agent state (SoA, env-major)
    action_u                     [B, N, 2]   raw physical action (scaled in-kernel)
    step_count                   [B] int32   MultiAgentEnv.current_step per env
    landmark_pos                 [B, L, 2]   landmark positions (visual / basic env)

`World.step()` launches the hand-written HIP kernel through the C ABI
(`fg_physics_step`).  Entity / Agent / Landmark objects are lightweight
descriptors (size, collide, movable, ...) whose `.state.p_pos` etc. are views of
the batched tensors, so scenario code written against the reference's names
keeps working.
"""
import numpy as np
import torch

from . import _native


# Every attribute write on a World / Entity / Wall bumps this counter: what depends on them (the FgParams a launch is bound
# with, MultiAgentEnv._bound_step) is re-derived when it has moved instead of being re-hashed on every step (ADVICE r3:
# params_signature() cost 40-160 us per step at 243-1024 agents, more than a small batch's kernel).
_version = [0]


class _Tracked(object):
    _UNTRACKED = frozenset()

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        if name not in self._UNTRACKED:
            _version[0] += 1


class EntityState(object):
    """Physical state of one entity across all B envs (core.py:4-9)."""

    def __init__(self, world=None, kind=None, index=None):
        self._world = world
        self._kind = kind
        self._index = index
        # Not bound to a device World (an entity of a host-side world description, e.g. the worlds a reference-style
        # Scenario file builds: formation_gym/callback_scenario.py): plain attributes, as in the reference (core.py:4-9)
        self._host = {}

    @property
    def p_pos(self):
        w, i = self._world, self._index
        if w is None:
            return self._host.get("p_pos")
        if self._kind == "agent":
            return torch.stack((w.pos_x[:, i], w.pos_y[:, i]), dim=-1)
        if self._kind == "obstacle":
            return w.obstacle_pos[:, i]
        return w.landmark_pos[:, i]

    @p_pos.setter
    def p_pos(self, value):
        w, i = self._world, self._index
        if w is None:
            self._host["p_pos"] = value
            return
        value = torch.as_tensor(value, dtype=torch.float32, device=w.device)
        w.state_version += 1
        if self._kind == "agent":
            w.pos_x[:, i] = value[..., 0]
            w.pos_y[:, i] = value[..., 1]
        elif self._kind == "obstacle":
            w.obstacle_pos[:, i] = value
        else:
            w.landmark_pos[:, i] = value

    @property
    def p_vel(self):
        w, i = self._world, self._index
        if w is None:
            return self._host.get("p_vel")
        if self._kind == "agent":
            return torch.stack((w.vel_x[:, i], w.vel_y[:, i]), dim=-1)
        if self._kind == "obstacle":
            return w.obstacle_vel[:, i]
        return torch.zeros((w.num_envs, 2), dtype=torch.float32, device=w.device)

    @p_vel.setter
    def p_vel(self, value):
        w, i = self._world, self._index
        if w is None:
            self._host["p_vel"] = value
            return
        value = torch.as_tensor(value, dtype=torch.float32, device=w.device)
        w.state_version += 1
        if self._kind == "obstacle":
            w.obstacle_vel[:, i] = value
            return
        if self._kind != "agent":
            return
        w.vel_x[:, i] = value[..., 0]
        w.vel_y[:, i] = value[..., 1]


class AgentState(EntityState):
    """Adds the communication utterance `c` (core.py:12-16): [B, dim_c], row `index` of `World.comm_c`.  All scenarios
    of the reference set `silent=True`, so `c` is identically zero there (core.py:281-282); a non-silent agent's `c` is
    what `World.step()` made of its `action.c` (core.py:284-286)."""

    @property
    def c(self):
        w = self._world
        if w is None:
            return self._host.get("c")
        if w.comm_c is not None:
            return w.comm_c[:, self._index]
        return torch.zeros((w.num_envs, w.dim_c), dtype=torch.float32, device=w.device)

    @c.setter
    def c(self, value):
        w = self._world
        if w is None:
            self._host["c"] = value
            return
        w.ensure_comm()[0][:, self._index] = torch.as_tensor(value, dtype=torch.float32, device=w.device)


class Action(object):
    """Physical action `u` and communication action `c` (core.py:19-24)."""

    def __init__(self, world=None, index=None):
        self._world = world
        self._index = index
        self._host = {}                   # an action of a host-side world description: plain attributes

    @property
    def c(self):
        """Communication action [B, dim_c] (row of `World.action_c`); None until somebody sets one."""
        w = self._world
        if w is None:
            return self._host.get("c")
        return None if w.action_c is None else w.action_c[:, self._index]

    @c.setter
    def c(self, value):
        w = self._world
        if w is None:
            self._host["c"] = value
            return
        if value is None:
            return
        w.ensure_comm()[1][:, self._index] = torch.as_tensor(value, dtype=torch.float32, device=w.device)

    @property
    def u(self):
        if self._world is None:
            return self._host.get("u")
        return self._world.action_u[:, self._index]

    @u.setter
    def u(self, value):
        w = self._world
        if w is None:
            self._host["u"] = value
            return
        w.action_u[:, self._index] = torch.as_tensor(value, dtype=torch.float32, device=w.device)


class Wall(_Tracked):
    """Wall description (core.py:27-41).  No reference scenario creates walls
    (world.walls == []); when present (at most 4) they are simulated in-kernel
    (core.py:325-362); a soft wall (hard=False) lets ghost entities through (:326-327)."""

    def __init__(self, orient="H", axis_pos=0.0, endpoints=(-1, 1), width=0.1, hard=True):
        self.orient = orient
        self.axis_pos = axis_pos
        self.endpoints = np.array(endpoints)
        self.width = width
        self.hard = hard
        self.color = np.array([0.0, 0.0, 0.0])


class Entity(_Tracked):
    """Properties of a physical world entity (core.py:45-75); identical for all B envs."""

    def __init__(self):
        self.i = 0
        self.name = ""
        self.size = 0.050
        self.movable = False
        self.collide = True
        self.ghost = False
        self.density = 25.0
        self.color = None
        self.max_speed = None
        self.accel = None
        self.state = EntityState()
        self.initial_mass = 1.0
        self.channel = None

    @property
    def mass(self):
        return self.initial_mass


class Landmark(Entity):
    def __init__(self):
        super(Landmark, self).__init__()


class Agent(Entity):
    """core.py:83-109."""

    def __init__(self):
        super(Agent, self).__init__()
        self.adversary = False
        self.dummy = False
        self.movable = True
        self.silent = False
        self.blind = False
        self.u_noise = None
        self.c_noise = None
        self.u_range = 1.0
        self.state = AgentState()
        self.action = Action()
        self.action_callback = None
        self.goal = None


class World(_Tracked):
    """B independent multi-agent worlds stepped in lock-step on one GPU.

    Constants and their reference defaults: core.py:113-139."""
    _UNTRACKED = frozenset(("world_step", "_props", "_sig_cache", "_silent_cache", "_frozen_cache", "_policy_cache", "state_version"))   # hot path / caches

    def __init__(self, world_length=50, num_envs=1, device=None):
        self.agents = []
        self.landmarks = []
        self.walls = []
        self.dim_c = 0
        self.dim_p = 2
        self.dim_color = 3
        self.dt = 0.1
        self.damping = 0.25
        self.contact_force = 1e+2
        self.contact_margin = 1e-3
        self.cache_dists = False
        self.world_length = world_length
        self.world_step = 0
        self.rng_counter = None          # device int64 [1]: see MultiAgentEnv.use_device_rng_counter
        self.num_agents = 0
        self.num_landmarks = 0
        # batched extension
        self.num_envs = int(num_envs)
        self.device = torch.device(device if device is not None else "cuda:0")
        self.pos_x = self.pos_y = self.vel_x = self.vel_y = None
        self.action_u = None
        self.step_count = None
        self.landmark_pos = None
        self.obstacle_pos = None          # movable colliding landmarks (formation_hd_obs_env)
        self.obstacle_vel = None
        self.comm_c = None                # AgentState.c of all agents [B, N, dim_c]; allocated with the first non-silent agent
        self.action_c = None              # Action.c [B, N, dim_c]
        self._props = None                # (signature, device table [N, 8]) of heterogeneous agents
        self.state_version = 0            # bumped by every host-side write of the device state (set_state, entity.state setters)
        self._sig_cache = None            # (key, params_signature()) - see _version
        self._silent_cache = None
        self.scenario = None

    # ---- reference-compatible views --------------------------------------
    @property
    def entities(self):
        return self.agents + self.landmarks

    @property
    def policy_agents(self):
        """core.py:152-154; entities the callback adapter simulates as physics-only bodies (colliding landmarks of a
        reference-style Scenario file, callback_scenario.py) are agents of the device World but nobody's policy."""
        # rebuilt only after an entity attribute was written or the agent list changed length (every env.step asks for it:
        # a list comprehension over 243 agents is 15 us of host time per step); a fresh list each time, as the reference returns
        key = (_version[0], len(self.agents))
        cache = getattr(self, "_policy_cache", None)
        if cache is None or cache[0] != key:
            cache = (key, [agent for agent in self.agents if agent.action_callback is None and not getattr(agent, "_physics_only", False)])
            self._policy_cache = cache
        return list(cache[1])

    @property
    def scripted_agents(self):
        return [agent for agent in self.agents if agent.action_callback is not None]

    # ---- storage ----------------------------------------------------------
    def allocate(self):
        """Create the SoA state tensors once agents/landmarks are defined."""
        if self.device.type != "cuda":
            raise RuntimeError("formation_gym (MI355X-native) needs a CUDA/HIP device; got %s. "
                               "There is no CPU fallback for the hot path." % self.device)
        _native.load()
        movable = [l for l in self.landmarks if l.movable]
        static = [l for l in self.landmarks if not l.movable]
        if self.landmarks != static + movable:
            raise ValueError("movable landmarks (obstacles) must come last in world.landmarks")
        B, N, L, M = self.num_envs, len(self.agents), len(static), len(movable)
        f = dict(dtype=torch.float32, device=self.device)
        self.pos_x = torch.zeros((B, N), **f)
        self.pos_y = torch.zeros((B, N), **f)
        self.vel_x = torch.zeros((B, N), **f)
        self.vel_y = torch.zeros((B, N), **f)
        self.action_u = torch.zeros((B, N, 2), **f)
        self.step_count = torch.zeros((B,), dtype=torch.int32, device=self.device)
        self.landmark_pos = torch.zeros((B, max(L, 1), 2), **f)
        self.obstacle_pos = torch.zeros((B, max(M, 1), 2), **f)
        self.obstacle_vel = torch.zeros((B, max(M, 1), 2), **f)
        self.num_agents, self.num_landmarks, self.num_obstacles = N, L, M
        for i, a in enumerate(self.agents):
            a.i = i
            a.state = AgentState(self, "agent", i)
            a.action = Action(self, i)
        for i, l in enumerate(self.landmarks):
            l.i = N + i
            l.state = EntityState(self, "landmark", i) if i < L else EntityState(self, "obstacle", i - L)

    def ensure_comm(self):
        """(comm_c, action_c): the communication state / action tensors [B, N, dim_c], allocated on first use."""
        if self.comm_c is None:
            if self.dim_c < 1:
                raise ValueError("non-silent agents need World.dim_c >= 1, got %d" % self.dim_c)
            f = dict(dtype=torch.float32, device=self.device)
            self.comm_c = torch.zeros((self.num_envs, len(self.agents), self.dim_c), **f)
            self.action_c = torch.zeros((self.num_envs, len(self.agents), self.dim_c), **f)
        return self.comm_c, self.action_c

    def any_non_silent(self):
        key = (_version[0], len(self.agents))
        if self._silent_cache is None or self._silent_cache[0] != key:
            self._silent_cache = (key, any(not a.silent for a in self.agents))
        return self._silent_cache[1]

    def any_frozen_silent(self):
        """Is some policy agent immovable and silent (the case `_set_action` asserts on, environment.py:191-236)?  Cached
        behind the entities' edit counter like `any_non_silent`: not a Python scan of every agent per step."""
        key = (_version[0], len(self.agents))
        cache = getattr(self, "_frozen_cache", None)
        if cache is None or cache[0] != key:
            cache = (key, any((not a.movable) and a.silent for a in self.policy_agents))
            self._frozen_cache = cache
        return cache[1]

    def agent_props(self):
        """Device table float [N, 8] = (mass, size, accel, max_speed, u_noise, c_noise, flags, 0) per agent for
        `FgParams.agent_props`, or None while all agents are alike (then the scalars of FgParams say it all).
        accel / max_speed / u_noise / c_noise: 0 = None; c_noise < 0 marks a silent agent (`fg_update_comm`);
        flags: 1 = not movable, 2 = does not collide, 4 = ghost (core.py:54-58), 0 for the agents of every reference scenario."""
        def flags(a):
            return float((0 if a.movable else _native.AGENT_IMMOVABLE) | (0 if a.collide else _native.AGENT_NO_COLLIDE) |
                         (_native.AGENT_GHOST if a.ghost else 0) |
                         (_native.AGENT_SCRIPTED if a.action_callback is not None else 0))
        rows = [(float(a.mass), float(a.size), float(a.accel or 0.0), float(a.max_speed or 0.0), float(a.u_noise or 0.0),
                 -1.0 if a.silent else float(a.c_noise or 0.0), flags(a), 0.0) for a in self.agents]
        alike = len({r[:5] + r[6:] for r in rows}) == 1 and rows[0][6] == 0.0      # any flag: the table (the scalars cannot say it)
        if alike and not self.any_non_silent():
            return None
        if alike and all(r[5] == rows[0][5] for r in rows) and rows[0][5] == 0.0:
            return None                                   # everybody non-silent, noise-free: the library's default
        sig = tuple(rows)
        if self._props is None or self._props[0] != sig:
            self._props = (sig, torch.tensor(rows, dtype=torch.float32, device=self.device), alike)
        return self._props[1]

    def set_state(self, pos=None, vel=None, mask=None):
        """Upload [B,N,2] positions / velocities (any array-like, or a tensor on any device) into the SoA tensors.
        mask: bool [B] on the device - only those envs take the new values (a device-side select: no synchronisation)."""
        def as_dev(x):
            if not torch.is_tensor(x):
                x = torch.as_tensor(np.asarray(x), dtype=torch.float32)
            return x.to(device=self.device, dtype=torch.float32)
        self.state_version += 1

        def put(dst, src):
            dst.copy_(src if mask is None else torch.where(mask[:, None], src, dst))
        if pos is not None:
            pos = as_dev(pos)
            put(self.pos_x, pos[..., 0]); put(self.pos_y, pos[..., 1])
        if vel is not None:
            vel = as_dev(vel)
            put(self.vel_x, vel[..., 0]); put(self.vel_y, vel[..., 1])

    def get_state(self):
        """(pos[B,N,2], vel[B,N,2]) as new tensors."""
        return (torch.stack((self.pos_x, self.pos_y), -1), torch.stack((self.vel_x, self.vel_y), -1))

    # ---- physics ------------------------------------------------------------
    def native_params(self, sensitivity=5.0, collide_thresh=0.0, auto_reset=False, seed=0, rng_offset=0, scripted_ok=False):
        """FgParams for the C ABI from this world's constants (uniform agents)."""
        a0 = self.agents[0]
        if self.scripted_agents and not scripted_ok:
            # core.py:210-211 runs `agent.action = agent.action_callback(agent, self)` inside World.step: `World.step()` here
            # evaluates a BATCHED callback on device tensors (see `run_scripted_agents`).  The fused env.step / rollout
            # launches compute rewards and observations for policy agents and have no place to call it between their steps.
            raise NotImplementedError("scripted agents (Agent.action_callback, core.py:210-211) are driven through the World "
                                      "API: World.step() calls the batched callback; env.step / rollout do not")
        # agents that differ in mass / size / accel / max_speed / u_noise (core.py:45-109): a per-agent table; the
        # scalars below then only carry agent 0's contact distance (the scale of collide_thresh) and the sensitivity
        # of agents without an accel of their own
        props = self.agent_props()
        hetero = props is not None and not self._props[2]
        if len(self.walls) > _native.MAX_WALLS:
            raise NotImplementedError("at most %d walls" % _native.MAX_WALLS)
        sens = a0.accel if (a0.accel is not None and not hetero) else sensitivity     # environment.py:218-220
        p = _native.FgParams(
            dt=self.dt, damping=self.damping, contact_force=self.contact_force,
            contact_margin=self.contact_margin, sensitivity=sens, mass=a0.mass,
            dist_min=a0.size + a0.size, collide_thresh=collide_thresh,
            world_length=int(self.world_length), auto_reset=1 if auto_reset else 0,
            seed=int(seed), rng_offset=int(rng_offset),
            accel=float(a0.accel or 0.0), max_speed=float(a0.max_speed or 0.0),
            u_noise=float(a0.u_noise or 0.0), num_walls=len(self.walls))
        counter = self.rng_counter               # MultiAgentEnv.use_device_rng_counter
        if counter is not None:
            p.rng_offset_dev = counter.data_ptr()
        if hetero:
            p.agent_props = props.data_ptr()
        for k, w in enumerate(self.walls):
            p.walls[k] = _native.FgWall(vertical=0 if w.orient == "H" else 1, axis_pos=float(w.axis_pos),
                                        end0=float(w.endpoints[0]), end1=float(w.endpoints[1]),
                                        width=float(w.width), soft=0 if w.hard else 1)
        return p

    def params_signature(self):
        """Tuple of everything `native_params` reads from the world and its agents: callers that cache an FgParams re-derive
        it when this changes.  Built again only after some attribute of the world, an entity or a wall was written (or the
        agent / wall lists changed length); the per-step cost is one small key comparison."""
        key = (_version[0], len(self.agents), len(self.walls), None if self.comm_c is None else self.comm_c.data_ptr(),
               None if self.rng_counter is None else self.rng_counter.data_ptr())
        if self._sig_cache is not None and self._sig_cache[0] == key:
            return self._sig_cache[1]
        sig = self._params_signature()
        self._sig_cache = (key, sig)
        return sig

    def _params_signature(self):
        return (self.dt, self.damping, self.contact_force, self.contact_margin, self.world_length,
                len(self.agents), tuple((a.size, a.initial_mass, a.accel, a.max_speed, a.u_noise, a.silent, a.c_noise,
                                         a.movable, a.collide, a.ghost, a.action_callback is not None) for a in self.agents),
                None if self.comm_c is None else self.comm_c.data_ptr(),
                None if self.rng_counter is None else self.rng_counter.data_ptr(),
                tuple((w.orient, float(w.axis_pos), float(w.endpoints[0]), float(w.endpoints[1]), float(w.width), w.hard)
                      for w in self.walls))

    def step(self, sensitivity=5.0, rng_offset=None):
        """World.step (core.py:206-225) for all envs: action force, all-pairs
        contact force, integration - one HIP launch.  `action_u` holds the RAW
        action; environment.py:216-221's sensitivity scaling happens in-kernel.
        rng_offset: the offset of the motor-noise draws (default: this world's own step count - fresh draws every step)."""
        self.world_step += 1
        self.run_scripted_agents()                        # core.py:210-211
        p = self.native_params(sensitivity=sensitivity, rng_offset=self.world_step if rng_offset is None else rng_offset,
                               scripted_ok=True)
        lib = _native.load()
        _native.check(lib.fg_physics_step(
            p, self.num_envs, len(self.agents),
            self.pos_x.data_ptr(), self.pos_y.data_ptr(), self.vel_x.data_ptr(), self.vel_y.data_ptr(),
            self.action_u.data_ptr(), _native.current_stream(self.device)))
        self.update_agent_state()
        if self.scenario is not None and hasattr(self.scenario, "_cache"):
            self.scenario._cache = None                   # per-agent callbacks must re-evaluate on the new state

    def run_scripted_agents(self):
        """core.py:210-211 `agent.action = agent.action_callback(agent, self)` for every scripted agent, BATCHED: the callback
        is called once per step with the Agent (whose `state.p_pos / p_vel` are [B, 2] device tensors) and this World, and
        returns the agent's action for all B envs - an object with `.u` (and optionally `.c`) or the `u` tensor itself,
        [B, 2] or broadcastable.  The values are used as they are: the sensitivity scaling of environment.py:216-221 applies
        to policy agents only (FG_AGENT_SCRIPTED in the agent's flags)."""
        for agent in self.scripted_agents:
            res = agent.action_callback(agent, self)
            u = getattr(res, "u", res)
            i = self.agents.index(agent)
            self.action_u[:, i] = torch.as_tensor(u, dtype=torch.float32, device=self.device).expand(self.num_envs, 2)
            c = getattr(res, "c", None)
            if c is not None and res is not agent.action and not agent.silent:
                agent.action.c = c

    def update_agent_state(self, seed=0):
        """core.py:221-222, 279-286 for every agent: `state.c` = zeros for a silent agent, `action.c` (+ c_noise) for
        the others - one launch (`fg_update_comm`); nothing to do while everybody is silent (`state.c` reads zeros)."""
        if not self.any_non_silent():
            if self.comm_c is not None:
                self.comm_c.zero_()
            return
        comm, act = self.ensure_comm()
        p = self.native_params(seed=seed, rng_offset=self.world_step, scripted_ok=True)
        props = self.agent_props()
        if props is not None:
            p.agent_props = props.data_ptr()              # the c_noise / silent column matters here even for alike agents
        _native.check(_native.load().fg_update_comm_dim(p, self.num_envs, len(self.agents), int(self.dim_c), act.data_ptr(),
                                                        comm.data_ptr(), _native.current_stream(self.device)))
