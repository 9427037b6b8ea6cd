"""Adapter for reference-style Scenario files (reference formation_gym/scenario.py:4-12, call sites
environment.py:113-184, __init__.py:6-17).

A scenario written for the reference knows nothing of batches or devices: it has `make_world(num_agents)`,
`reset_world(world)`, `observation(agent, world)`, `reward(agent, world)` (+ optionally `benchmark_data`), builds its
world from `formation_gym.core.World / Agent / Landmark`, and reads `entity.state.p_pos` etc. as NumPy vectors.  The
kernels of this package cannot run arbitrary Python, so such a file takes the SLOW path, loudly labelled
(`env.info['path']`, one warning): `_set_action` and `World.step` - the O(N^2) physics - run on the GPU for all envs in
one launch (`fg_physics_step`, including agents of different mass / size / accel / max_speed, walls, and LANDMARKS THAT
COLLIDE - movable ones like the reference's formation_hd_obs_env obstacles, immovable ones that only push back: they ride
along as physics-only bodies behind the agents, core.py:240-262 runs over all entities), the state comes
back to the host once per step, and the file's own per-agent callbacks run there on host views of it, exactly as
the reference calls them (environment.py:126-134).  Scenarios that want the fused path implement the batched protocol of
`scenario.BaseScenario` (step_batch / observe_batch), as the five scenarios in `envs/` do.

One instance of the user's Scenario per environment (its attributes - a target shape, say - belong to one env), each with
its own host-side world description and its own legacy NumPy stream: env b draws its resets from
`np.random.seed(seed + 1000 (env_base + b))`, swapped into NumPy's GLOBAL generator around the user's `reset_world`, which
is what the reference's `env.seed` + worker convention amounts to (environment.py:106-110, train/maddpg-v2/main.py:19-30).
"""
import warnings

import numpy as np
import torch

from .core import Agent, Landmark, World
from .scenario import BaseScenario


class CallbackScenario(BaseScenario):
    PATH = ("host callbacks (reference-style Scenario file): _set_action + World.step on the GPU, "
            "observation / reward / benchmark_data per agent in Python on the host")

    def __init__(self, factory):
        self._factory = factory           # () -> a fresh instance of the user's Scenario class
        self.users = []                   # one per env
        self.host_worlds = []
        self._seed = 1
        self._rng_states = None
        self._cache = None
        self._step_host = None
        self._uploads = 0                 # bumped by every host -> device copy: with the step counters it stamps a device state
        self._downloaded = None

    # ---- construction -------------------------------------------------------
    def make_world(self, num_agents=3, num_envs=1, device=None, **kwargs):
        B = int(num_envs)
        self.users = [self._factory() for _ in range(max(B, 1))]
        saved = np.random.get_state()
        try:
            self.host_worlds = [u.make_world(num_agents, **kwargs) for u in self.users]
        finally:
            np.random.set_state(saved)    # the draws of make_world's own reset_world must not move the caller's stream
        hw = self.host_worlds[0]
        world = World(world_length=getattr(hw, "world_length", 50), num_envs=B, device=device)
        for k in ("dim_c", "dim_p", "dt", "damping", "contact_force", "contact_margin", "collaborative", "discrete_action"):
            if hasattr(hw, k):
                setattr(world, k, getattr(hw, k))
        world.walls = list(hw.walls)
        # Landmarks that take part in the physics (core.py:292-295: collide, and somebody of the pair moves) are simulated as
        # bodies BEHIND the agents in the device World's entity table, in the reference's entity order (agents, then
        # landmarks in list order: core.py:147-149), flagged immovable where they are; the others stay host-side.
        self._bodies = [i for i, l in enumerate(hw.landmarks)
                        if l.movable or (l.collide and any(a.collide for a in hw.agents))]     # (a movable non-collider still drifts, :264-277)
        self._statics = [i for i in range(len(hw.landmarks)) if i not in self._bodies]
        self._num_real = len(hw.agents)
        bodies = []
        for i in self._bodies:
            b_ = self._clone(hw.landmarks[i], Agent())
            b_.silent = True; b_._physics_only = True
            b_.accel = None; b_.u_noise = None; b_.c_noise = None
            bodies.append(b_)
        world.agents = [self._clone(a, Agent()) for a in hw.agents] + bodies
        world.landmarks = [self._clone(hw.landmarks[i], Landmark()) for i in self._statics]
        world.allocate()
        world.scenario = self
        self._step_host = np.zeros(B, dtype=np.int64)
        self._upload(world)
        warnings.warn("formation_gym: %s - %s" % (type(self.users[0]).__module__, self.PATH), stacklevel=3)
        return world

    @staticmethod
    def _clone(src, dst):
        for k, v in vars(src).items():
            if k not in ("state", "action"):
                setattr(dst, k, v)
        return dst

    # ---- host <-> device ----------------------------------------------------
    def _upload(self, world):
        """Host world descriptions -> the device World's SoA tensors."""
        B, N = world.num_envs, len(world.agents)
        pos = np.zeros((B, N, 2)); vel = np.zeros((B, N, 2))
        lm = np.zeros((B, max(len(world.landmarks), 1), 2))
        for b in range(B):
            hw = self.host_worlds[b]
            for i, a in enumerate(self._host_entities(hw)):
                pos[b, i] = a.state.p_pos
                vel[b, i] = a.state.p_vel if a.state.p_vel is not None else 0.0
            for k, i in enumerate(self._statics):
                lm[b, k] = hw.landmarks[i].state.p_pos
        world.set_state(pos, vel)
        world.landmark_pos.copy_(torch.as_tensor(lm, dtype=torch.float32))
        self._cache = None
        self._uploads += 1

    def _host_entities(self, hw):
        """The host world's simulated entities in device order: its agents, then its colliding landmarks."""
        return list(hw.agents) + [hw.landmarks[i] for i in self._bodies]

    def _download(self, world):
        """Device state -> the host worlds' entity states (float64 views of the fp32 values).  Once per device state: the
        per-agent wrappers (observation / reward / benchmark_data for one agent after the other) share the copy (ADVICE r3:
        each of them used to download the whole state again)."""
        stamp = (world.world_step, world.state_version, int(self._step_host.sum()) if self._step_host is not None else 0, self._uploads)
        if self._downloaded == stamp:
            return
        self._downloaded = stamp
        pos, vel = world.get_state()
        pos = pos.double().cpu().numpy(); vel = vel.double().cpu().numpy()
        for b, hw in enumerate(self.host_worlds[:world.num_envs]):
            for i, a in enumerate(self._host_entities(hw)):
                a.state.p_pos = pos[b, i].copy()
                a.state.p_vel = vel[b, i].copy()
            for a in hw.agents:
                if a.state.c is None or a.silent:
                    a.state.c = np.zeros(hw.dim_c)

    def _upload_bodies(self, world):
        """Host -> device for the colliding landmarks only: a reward callback may have written their state (the reference's
        formation_hd_obs_env re-arms its obstacles' velocity there, :82-89)."""
        if not self._bodies:
            return
        n0, nb = self._num_real, len(self._bodies)
        B = world.num_envs
        pos = np.zeros((B, nb, 2), dtype=np.float32); vel = np.zeros((B, nb, 2), dtype=np.float32)
        for b in range(B):
            for k, i in enumerate(self._bodies):
                st = self.host_worlds[b].landmarks[i].state
                pos[b, k] = st.p_pos
                vel[b, k] = st.p_vel if st.p_vel is not None else 0.0
        dev = world.device
        world.pos_x[:, n0:] = torch.as_tensor(pos[..., 0]).to(dev); world.pos_y[:, n0:] = torch.as_tensor(pos[..., 1]).to(dev)
        world.vel_x[:, n0:] = torch.as_tensor(vel[..., 0]).to(dev); world.vel_y[:, n0:] = torch.as_tensor(vel[..., 1]).to(dev)
        self._uploads += 1

    # ---- RNG: the reference draws from NumPy's global legacy generator -------
    def seed(self, seed=None):
        self._seed = 1 if seed is None else int(seed)
        self._rng_states = None

    def _with_stream(self, b, fn):
        if self._rng_states is None:
            base = getattr(self, "env_base", 0)
            self._rng_states = [np.random.RandomState(self._seed + 1000 * (base + k)).get_state()
                                for k in range(len(self.users))]
        saved = np.random.get_state()
        np.random.set_state(self._rng_states[b])
        try:
            return fn()
        finally:
            self._rng_states[b] = np.random.get_state()
            np.random.set_state(saved)

    def reset_world(self, world, env_mask=None):
        for b in range(world.num_envs):
            if env_mask is None or env_mask[b]:
                self._with_stream(b, lambda b=b: self.users[b].reset_world(self.host_worlds[b]))
                self._step_host[b] = 0
        if env_mask is None:
            self._upload(world)
            world.step_count.zero_()
        else:                                     # envs that keep running keep their device state
            self._download_masked(world, env_mask)
            self._upload(world)
            m = torch.as_tensor(np.asarray(env_mask, dtype=bool), device=world.device)
            world.step_count.masked_fill_(m, 0)

    def _download_masked(self, world, env_mask):
        pos, vel = world.get_state()
        pos = pos.double().cpu().numpy(); vel = vel.double().cpu().numpy()
        for b, hw in enumerate(self.host_worlds[:world.num_envs]):
            if env_mask[b]:
                continue
            for i, a in enumerate(self._host_entities(hw)):
                a.state.p_pos = pos[b, i].copy(); a.state.p_vel = vel[b, i].copy()

    # ---- batched protocol (what MultiAgentEnv drives) ------------------------
    def obs_dim(self, world):
        hw = self.host_worlds[0]
        return int(len(self.users[0].observation(hw.agents[0], hw)))

    def _callbacks(self, world, out, rewards=True):
        """environment.py:126-134 for every env: the user's observation / reward per agent, on the host.  rewards=False:
        what reset() does (environment.py:154-155: observations only - a reward callback with side effects must not run)."""
        B, N = world.num_envs, self._num_real
        self._download(world)
        D = out["obs"].shape[-1]
        obs = np.zeros((B, N, D), dtype=np.float32)
        indiv = np.zeros((B, N), dtype=np.float32)
        first = np.zeros((B, N), dtype=np.float32)
        for b in range(B):
            u, hw = self.users[b], self.host_worlds[b]
            for i, a in enumerate(hw.agents):
                # environment.py:127-130: observation, reward for reward_n, and the reward AGAIN for info['individual_reward']
                obs[b, i] = np.asarray(u.observation(a, hw), dtype=np.float64)
                if rewards:
                    first[b, i] = float(u.reward(a, hw))        # :128 reward_n, :136 its sum = the shared reward
                    indiv[b, i] = float(u.reward(a, hw))        # :130 info['individual_reward'] - a second call
        if rewards:
            self._upload_bodies(world)
        out["obs"].copy_(torch.as_tensor(obs))
        if out.get("indiv") is not None:
            out["indiv"].copy_(torch.as_tensor(indiv))
        if out.get("reward") is not None:       # environment.py:136-138: the shared reward is the sum over agents' FIRST values
            shared = first.astype(np.float64).sum(1, keepdims=True).astype(np.float32)
            out["reward"].copy_(torch.as_tensor(np.repeat(shared, N, 1)))
        # what reward_n holds when the reward is not shared: the first call's values (a reward callback that is not idempotent
        # returns something else the second time; MultiAgentEnv reads this instead of out["indiv"] then)
        self.first_reward = torch.as_tensor(first).to(out["obs"].device) if rewards else None
        if out.get("done") is not None:         # environment.py:172-178
            done = (self._step_host[:B] >= int(world.world_length)).astype(np.uint8)
            out["done"].copy_(torch.as_tensor(np.repeat(done[:, None], N, 1)))
        self._cache = out

    def step_batch(self, world, act, out, auto_reset=False, rng_offset=0):
        if self._bodies:                          # the bodies behind the agents take no action (core.py:229-237: agents only)
            world.action_u.zero_()
            world.action_u[:, :self._num_real].copy_(act)
        elif act.data_ptr() != world.action_u.data_ptr():
            world.action_u.copy_(act)
        world.step()                              # _set_action's scaling + World.step: one launch (fg_physics_step)
        world.world_step -= 1                     # MultiAgentEnv.step counts the step itself
        world.step_count.add_(1)
        self._step_host += 1
        self._callbacks(world, out)
        if auto_reset:
            # the vec-env worker's rule (env_wrappers.py:14-18) on the host, where this scenario's callbacks live: an env whose
            # agents are all done restarts from its own stream and the RESET observation goes out with the finished step's
            # reward / done (what FormationVecEnv's default reset mode asks of every scenario)
            finished = self._step_host[:world.num_envs] >= int(world.world_length)
            if finished.any():
                keep = {k: out[k].clone() for k in ("reward", "indiv", "done") if out.get(k) is not None}
                first = self.first_reward
                self.reset_world(world, env_mask=finished)
                self._callbacks(world, {"obs": out["obs"]}, rewards=False)
                for k, v in keep.items():
                    out[k].copy_(v)
                self.first_reward = first
                self._cache = None

    def observe_batch(self, world, out):
        self._callbacks(world, out, rewards=False)

    # ---- per-agent callbacks with the reference's signature ------------------
    def _host_agent(self, agent, b=0):
        return self.host_worlds[b].agents[agent.i]

    def observation(self, agent, world):
        hw = self.host_worlds[0]
        self._download(world)
        return self.users[0].observation(self._host_agent(agent), hw)

    def reward(self, agent, world):
        hw = self.host_worlds[0]
        self._download(world)
        return self.users[0].reward(self._host_agent(agent), hw)

    def benchmark_data(self, agent, world):
        fn = getattr(self.users[0], "benchmark_data", None)
        if fn is None:
            return {}
        self._download(world)
        return fn(self._host_agent(agent), self.host_worlds[0])

    def info(self, agent, world):
        fn = getattr(self.users[0], "info", None)
        return fn(self._host_agent(agent), self.host_worlds[0]) if fn else {}
