"""basic_formation_env: MPE simple_spread-style task (reference
formation_gym/envs/basic_formation_env.py), MI355X-native.

The reference file imports OpenAI's `multiagent` package (:3-4), which it does
not vendor; this plugin uses this package's own core types.  Observation
(:29-41), reward (:43-52, self-"collision" included) and done run in the HIP
kernel `fg_step_basic`.
"""
import numpy as np
import torch

from formation_gym import _native
from formation_gym.core import World, Agent, Landmark
from formation_gym.scenario import BaseScenario
from formation_gym.landmark_scenario import MtResetMixin


class Scenario(MtResetMixin, BaseScenario):
    def make_world(self, num_agents=3, num_landmarks=3, num_envs=1, device=None):
        world = World(num_envs=num_envs, device=device)      # world_length = 50 (core.py:113)
        world.dim_c = 2
        world.collaborative = True
        world.agents = [Agent() for _ in range(num_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = True
            agent.silent = True
            agent.size = 0.1
        world.landmarks = [Landmark() for _ in range(num_landmarks)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmarks %d' % i
            landmark.collide = False
            landmark.movable = False
        world.allocate()
        world.scenario = self
        self._rngs = None
        self._seed = 1
        self._cache = None
        self.reset_world(world)
        return world

    def seed(self, seed=None):
        self._seed = 1 if seed is None else int(seed)
        self._rngs = None

    def _streams(self, B):
        if self._rngs is None or len(self._rngs) != B:
            self._rngs = [np.random.RandomState(self._seed + 1000 * (getattr(self, "env_base", 0) + b)) for b in range(B)]
        return self._rngs

    def reset_world(self, world, env_mask=None):
        """:54-65 - agent positions then landmark positions, U(-1,1)^2."""
        B, N, L = world.num_envs, len(world.agents), len(world.landmarks)
        rngs = self._streams(B)
        pos, vel = world.get_state()
        pos = pos.cpu().numpy().astype(np.float64); vel = vel.cpu().numpy().astype(np.float64)
        lm = world.landmark_pos.cpu().numpy().astype(np.float64)
        for b in (range(B) if env_mask is None else [b for b in range(B) if env_mask[b]]):
            pos[b] = rngs[b].uniform(-1, +1, (N, 2))
            vel[b] = 0.0
            lm[b] = rngs[b].uniform(-1, +1, (L, 2))
        world.set_state(pos, vel)
        world.landmark_pos.copy_(torch.as_tensor(lm, dtype=torch.float32))
        if env_mask is None:
            world.step_count.zero_()
        else:
            world.step_count.masked_fill_(torch.as_tensor(np.asarray(env_mask, dtype=bool),
                                                          device=world.device), 0)
        self._cache = None

    def obs_dim(self, world):
        N, L = len(world.agents), len(world.landmarks)
        return 4 + 2 * L + 4 * (N - 1)

    def params(self, world, rng_offset=0, auto_reset=False):
        a0 = world.agents[0]
        # agents of different mass / size / accel / max_speed / u_noise and immovable, non-colliding or ghost agents
        # (core.py:45-109, 54-58) travel in World.agent_props(): the run-time-count kernel honours every column
        p = world.native_params(collide_thresh=a0.size + a0.size, auto_reset=auto_reset, seed=self._seed,
                                rng_offset=rng_offset)                                # :91
        # the device counter RNG (motor noise) is keyed by seed, GLOBAL env index and the per-step offset, like formation_hd_env's
        p.env_index_base = int(getattr(self, "env_base", 0))
        return p

    def _launch(self, world, act, out, do_physics, rng_offset=0, auto_reset=False):
        lib = _native.load()
        _native.check(lib.fg_step_basic(
            self.params(world, rng_offset, auto_reset), world.num_envs, len(world.agents), len(world.landmarks),
            1 if do_physics else 0,
            world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
            _native.ptr(act), world.landmark_pos.data_ptr(), world.step_count.data_ptr(),
            out["obs"].data_ptr(), _native.ptr(out.get("reward")), _native.ptr(out.get("indiv")),
            _native.ptr(out.get("done")), _native.ptr(out.get("near_ag")), _native.current_stream(world.device)))
        self._cache = out

    def bind_step(self, world, act, out, auto_reset=False):
        """Resolve FgParams and every pointer once; returns `launch(rng_offset)` (one ctypes call
        per step, see MultiAgentEnv._bound_step)."""
        lib = _native.load()
        p = self.params(world, auto_reset=auto_reset)
        args = (world.num_envs, len(world.agents), len(world.landmarks), 1,
                world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
                act.data_ptr(), world.landmark_pos.data_ptr(), world.step_count.data_ptr(),
                out["obs"].data_ptr(), _native.ptr(out.get("reward")), _native.ptr(out.get("indiv")),
                _native.ptr(out.get("done")), _native.ptr(out.get("near_ag")), _native.current_stream(world.device))
        fn = lib.fg_step_basic
        keep = (act, out)

        def launch(rng_offset=0):
            p.rng_offset = rng_offset
            rc = fn(p, *args)
            if rc:
                _native.check(rc)
            return keep
        self._cache = None
        return launch

    def step_batch(self, world, act, out, auto_reset=False, rng_offset=0):
        """auto_reset: envs whose episode ends restart inside the launch (device counter RNG; the reset observation comes
        back with the finished step's reward / done: the vec-env worker's rule, env_wrappers.py:14-18)."""
        self._launch(world, act, out, True, rng_offset, auto_reset)

    def _descriptor(self, world):
        return _native.FgScenario(kind=_native.FG_SCN_BASIC, num_landmarks=len(world.landmarks), num_obstacles=0, penalty=1.0,
                                  variant=int(getattr(self, "kernel_variant", 0)))   # 1: the run-time-count kernel (A/B runs)

    def _rollout_args(self, world, act_seq, out, obs_every):
        if out.get("obs") is not None and not out["obs"].is_contiguous():
            raise ValueError("basic_formation_env writes contiguous observation buffers")
        return (world.num_envs, len(world.agents), int(act_seq.shape[0]),
                world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
                act_seq.data_ptr(), world.landmark_pos.data_ptr(), None, None, world.step_count.data_ptr(),
                out["obs"].data_ptr(), out["reward"].data_ptr(), _native.ptr(out.get("indiv")),
                _native.ptr(out.get("done")), _native.ptr(out.get("near_ag")), int(obs_every),
                _native.current_stream(world.device))

    def rollout_batch(self, world, act_seq, out, obs_every=1, auto_reset=False, rng_offset=0):
        """K steps in one launch (`fg_rollout_scenario`); act_seq [K,B,N,2], out tensors carry a leading K."""
        _native.check(_native.load().fg_rollout_scenario(self.params(world, rng_offset, auto_reset), self._descriptor(world),
                                                         *self._rollout_args(world, act_seq, out, obs_every)))
        self._cache = None

    def bind_rollout(self, world, act_seq, out, obs_every=1, auto_reset=False):
        """`rollout_batch` with the structs and every pointer resolved once: returns `launch(rng_offset)`."""
        fn = _native.load().fg_rollout_scenario
        p, d = self.params(world, auto_reset=auto_reset), self._descriptor(world)
        args = self._rollout_args(world, act_seq, out, obs_every)
        keep = (act_seq, out)

        def launch(rng_offset=0):
            p.rng_offset = rng_offset
            rc = fn(p, d, *args)
            if rc:
                _native.check(rc)
            return keep
        return launch

    def reset_device(self, world, mask=None, rng_offset=0):
        """Throughput-mode reset on the GPU (counter RNG, distributional parity only): the draws the fused auto-reset makes."""
        _native.check(_native.load().fg_reset_scenario(
            self.params(world, rng_offset), self._descriptor(world), world.num_envs, len(world.agents), _native.ptr(mask),
            world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
            world.landmark_pos.data_ptr(), None, None, world.step_count.data_ptr(), _native.current_stream(world.device)))
        self._cache = None

    def observe_batch(self, world, out):
        self._launch(world, None, out, False)

    def _fresh(self, world):
        if self._cache is None:
            B, N = world.num_envs, len(world.agents)
            f = dict(dtype=torch.float32, device=world.device)
            out = dict(obs=torch.empty((B, N, self.obs_dim(world)), **f),
                       reward=torch.empty((B, N), **f), indiv=torch.empty((B, N), **f))
            self.observe_batch(world, out)
        return self._cache

    def observation(self, agent, world):
        return self._fresh(world)["obs"][:, agent.i]

    def reward(self, agent, world):
        return self._fresh(world)["indiv"][:, agent.i]

    def benchmark_data(self, agent, world):
        """:67-87."""
        pos, _ = world.get_state()
        d = (pos - pos[:, agent.i:agent.i + 1]).norm(dim=-1)
        dl = (pos[:, :, None, :] - world.landmark_pos[:, None, :, :]).norm(dim=-1).min(1).values
        return {'reward': self.reward(agent, world),
                'collisions': (d < 2 * world.agents[0].size).sum(1),
                'min_dists': dl.sum(1), 'occupied_landmarks': (dl < 0.1).sum(1)}

    def is_collision(self, agent1, agent2):
        dist = (agent1.state.p_pos - agent2.state.p_pos).norm(dim=-1)
        return dist < (agent1.size + agent2.size)
