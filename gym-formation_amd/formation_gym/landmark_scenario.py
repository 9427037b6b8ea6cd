"""Shared machinery of the landmark-formation scenarios with few agents
(formation_hd_partial_env, formation_hd_partial_range_env, formation_hd_obs_env):
agents chase a static set of landmarks, reward = -Hausdorff(centred agents,
centred landmarks) - collision penalties.  Observation, reward, done and the
physics (incl. movable colliding obstacles) run in the HIP kernel
`fg_step_scenario`; each scenario file supplies its constants."""
import numpy as np
import torch

from formation_gym import _native
from formation_gym.core import World, Agent, Landmark
from formation_gym.scenario import BaseScenario


class MtResetMixin(object):
    """`reset_mode='device_mt'` for the scenarios with landmarks: each env's legacy MT19937 stream (RandomState(seed + 1000 g),
    at its current position) continued on the GPU by `fg_reset_scenario_mt` - the draws of the host `reset_world`, bit for bit
    (basic_formation_env.py:54-65, formation_hd_partial_env.py:88-99, formation_hd_partial_range_env.py:76-87,
    formation_hd_obs_env.py:101-120).  The using class supplies `_streams(B)` and, for obstacles, `descriptor()`."""
    _mt_state = None

    def _mt_descriptor(self, world):
        if hasattr(self, "descriptor"):
            return self.descriptor()
        return _native.FgScenario(kind=_native.FG_SCN_BASIC, num_landmarks=len(world.landmarks), num_obstacles=0)

    def upload_mt_streams(self, world):
        B = world.num_envs
        st = np.zeros((B, 626), dtype=np.uint32)
        for b, rs in enumerate(self._streams(B)):
            _, key, pos = rs.get_state()[:3]
            st[b, :624] = key
            st[b, 624] = pos
        self._mt_state = torch.as_tensor(st.view(np.int32)).to(world.device)
        return self._mt_state

    def _mt_args(self, world, mask, world_length):
        if self._mt_state is None or self._mt_state.shape[0] != world.num_envs:
            self.upload_mt_streams(world)
        return (self._mt_descriptor(world), world.num_envs, len(world.agents), _native.ptr(mask), int(world_length),
                self._mt_state.data_ptr(), world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(),
                world.vel_y.data_ptr(), world.landmark_pos.data_ptr(), world.obstacle_pos.data_ptr(),
                world.obstacle_vel.data_ptr(), world.step_count.data_ptr(), _native.current_stream(world.device))

    def reset_mt(self, world, mask=None):
        """Scenario.reset_world on the GPU for the masked envs (mask None: every env)."""
        _native.check(_native.load().fg_reset_scenario_mt(*self._mt_args(world, mask, 0)))
        self._cache = None

    def bind_reset_mt_done(self, world, obs=None):
        """The vec-env worker's `if all(done): ob = env.reset()` (env_wrappers.py:14-18) decided on the device: envs whose step
        counter has reached world_length restart from their own streams; with `obs` the observation tensor is then brought up
        to date for the whole batch (one more launch: unchanged envs get the bits they had)."""
        lib = _native.load()
        args = self._mt_args(world, None, int(world.world_length))
        keep = (self._mt_state, obs)

        def launch():
            rc = lib.fg_reset_scenario_mt(*args)
            if rc:
                _native.check(rc)
            self._cache = None
            if obs is not None:
                self.observe_batch(world, {"obs": obs})
            return keep
        return launch

    def reset_mt_done(self, world, obs=None):
        self.bind_reset_mt_done(world, obs)()


class LandmarkScenario(MtResetMixin, BaseScenario):
    KIND = None                    # _native.FG_SCN_*
    AGENT_SIZE = 0.04
    LANDMARK_SIZE = 0.02
    OBSTACLE_SIZE = 0.15
    PENALTY = 1.0
    OBSTACLE_VEL = (0.0, -1.0)
    OBSTACLE_FLOOR = -2.2

    def _build_world(self, num_agents, num_landmarks, num_obstacles, world_length, num_envs, device):
        self.num_agents, self.num_landmarks, self.num_obstacles = num_agents, num_landmarks, num_obstacles
        world = World(num_envs=num_envs, device=device)
        world.world_length = world_length
        world.dim_c = 2
        world.collaborative = True
        world.agents = [Agent() for _ in range(num_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = True
            agent.silent = True
            agent.size = self.AGENT_SIZE
        world.landmarks = [Landmark() for _ in range(num_landmarks + num_obstacles)]
        for i, landmark in enumerate(world.landmarks):
            if i < num_landmarks:
                landmark.name = 'landmarks %d' % i
                landmark.collide = False
                landmark.movable = False
                landmark.size = self.LANDMARK_SIZE
            else:
                landmark.name = 'obstacles %d' % (i - num_landmarks)
                landmark.collide = True
                landmark.movable = True
                landmark.size = self.OBSTACLE_SIZE
        world.allocate()
        world.scenario = self
        self._rngs = None
        self._seed = 1
        self._cache = None
        self.num_obs = getattr(self, "num_obs", 0)
        self.obs_range = getattr(self, "obs_range", 0.0)
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
        """Agents then landmarks U(-1,1)^2; obstacles from U([step_k, 2.0], [step_k+1, 2.5]) with
        velocity (0,-1) (formation_hd_obs_env.py:101-114); one legacy MT19937 stream per env."""
        B, N, L, M = world.num_envs, len(world.agents), self.num_landmarks, self.num_obstacles
        rngs = self._streams(B)
        pos, vel = world.get_state()
        pos = pos.cpu().numpy().astype(np.float64); vel = vel.cpu().numpy().astype(np.float64)
        lm = world.landmark_pos.cpu().numpy().astype(np.float64)
        op = world.obstacle_pos.cpu().numpy().astype(np.float64)
        ov = world.obstacle_vel.cpu().numpy().astype(np.float64)
        step = np.linspace(-1.8, 1.8, M + 1) if M else None
        for b in (range(B) if env_mask is None else [b for b in range(B) if env_mask[b]]):
            rs = rngs[b]
            pos[b] = rs.uniform(-1, +1, (N, 2))
            vel[b] = 0.0
            for i in range(L + M):
                if i < L:
                    lm[b, i] = rs.uniform(-1, +1, 2)
                else:
                    k = i - L
                    op[b, k] = rs.uniform([step[k], 2.0], [step[k + 1], 2.5])
                    ov[b, k] = self.OBSTACLE_VEL
        world.set_state(pos, vel)
        world.landmark_pos.copy_(torch.as_tensor(lm, dtype=torch.float32))
        world.obstacle_pos.copy_(torch.as_tensor(op, dtype=torch.float32))
        world.obstacle_vel.copy_(torch.as_tensor(ov, dtype=torch.float32))
        if env_mask is None:
            world.step_count.zero_()
        else:
            world.step_count.masked_fill_(torch.as_tensor(np.asarray(env_mask, dtype=bool),
                                                          device=world.device), 0)
        self._cache = None

    def obs_dim(self, world):
        N, L, M = len(world.agents), self.num_landmarks, self.num_obstacles
        nbr = self.num_obs if self.KIND == _native.FG_SCN_PARTIAL else N - 1
        return 2 + 2 * L + 2 * M + 2 * nbr + 2 * (N - 1)

    def params(self, world, rng_offset=0, auto_reset=False):
        a0 = world.agents[0]
        # agents of different mass / size / accel / max_speed / u_noise and immovable, non-colliding or ghost agents
        # (core.py:45-109, 54-58) travel in World.agent_props(): the run-time-count kernel honours every column
        p = world.native_params(collide_thresh=a0.size + a0.size, auto_reset=auto_reset, seed=self._seed,
                                rng_offset=rng_offset)                                 # is_collision: size_a + size_b
        # the device counter RNG (motor noise) is keyed by seed, GLOBAL env index and the per-step offset, like formation_hd_env's
        p.env_index_base = int(getattr(self, "env_base", 0))
        return p

    def descriptor(self):
        return _native.FgScenario(kind=self.KIND, num_landmarks=self.num_landmarks,
                                  num_obstacles=self.num_obstacles, num_obs=int(self.num_obs),
                                  obs_range=float(self.obs_range), obstacle_size=self.OBSTACLE_SIZE,
                                  obstacle_vx=self.OBSTACLE_VEL[0], obstacle_vy=self.OBSTACLE_VEL[1],
                                  obstacle_floor=self.OBSTACLE_FLOOR, penalty=self.PENALTY,
                                  variant=int(getattr(self, "kernel_variant", 0)))   # 1: the run-time-count kernel (A/B runs)

    def _launch(self, world, act, out, do_physics, rng_offset=0, auto_reset=False):
        lib = _native.load()
        M = self.num_obstacles
        _native.check(lib.fg_step_scenario(
            self.params(world, rng_offset, auto_reset), self.descriptor(), world.num_envs, len(world.agents), 1 if do_physics else 0,
            world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
            _native.ptr(act), world.landmark_pos.data_ptr(),
            world.obstacle_pos.data_ptr() if M else None, world.obstacle_vel.data_ptr() if M else None,
            world.step_count.data_ptr(),
            out["obs"].data_ptr(), _native.ptr(out.get("reward")), _native.ptr(out.get("indiv")),
            _native.ptr(out.get("done")), _native.current_stream(world.device)))
        self._cache = out

    def bind_step(self, world, act, out, auto_reset=False):
        """Resolve the structs and every pointer once; returns `launch(rng_offset)` (one ctypes call
        per step, see MultiAgentEnv._bound_step)."""
        lib = _native.load()
        M = self.num_obstacles
        p, d = self.params(world, auto_reset=auto_reset), self.descriptor()
        args = (world.num_envs, len(world.agents), 1,
                world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
                act.data_ptr(), world.landmark_pos.data_ptr(),
                world.obstacle_pos.data_ptr() if M else None, world.obstacle_vel.data_ptr() if M else None,
                world.step_count.data_ptr(),
                out["obs"].data_ptr(), _native.ptr(out.get("reward")), _native.ptr(out.get("indiv")),
                _native.ptr(out.get("done")), _native.current_stream(world.device))
        fn = lib.fg_step_scenario
        keep = (act, out)

        def launch(rng_offset=0):
            p.rng_offset = rng_offset
            rc = fn(p, d, *args)
            if rc:
                _native.check(rc)
            return keep
        self._cache = None
        return launch

    def step_batch(self, world, act, out, auto_reset=False, rng_offset=0):
        """auto_reset: envs whose episode ends restart inside the launch (device counter RNG; the reset observation comes
        back with the finished step's reward / done: the vec-env worker's rule, env_wrappers.py:14-18)."""
        self._launch(world, act, out, True, rng_offset, auto_reset)

    def _rollout_args(self, world, act_seq, out, obs_every):
        M = self.num_obstacles
        if out.get("obs") is not None and not out["obs"].is_contiguous():
            raise ValueError("the landmark scenarios write contiguous observation buffers")
        return (world.num_envs, len(world.agents), int(act_seq.shape[0]),
                world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
                act_seq.data_ptr(), world.landmark_pos.data_ptr(),
                world.obstacle_pos.data_ptr() if M else None, world.obstacle_vel.data_ptr() if M else None,
                world.step_count.data_ptr(),
                out["obs"].data_ptr(), out["reward"].data_ptr(), _native.ptr(out.get("indiv")),
                _native.ptr(out.get("done")), None, int(obs_every), _native.current_stream(world.device))

    def rollout_batch(self, world, act_seq, out, obs_every=1, auto_reset=False, rng_offset=0):
        """K steps in one launch (`fg_rollout_scenario`); act_seq [K,B,N,2], out tensors carry a leading K."""
        _native.check(_native.load().fg_rollout_scenario(self.params(world, rng_offset, auto_reset), self.descriptor(),
                                                         *self._rollout_args(world, act_seq, out, obs_every)))
        self._cache = None

    def bind_rollout(self, world, act_seq, out, obs_every=1, auto_reset=False):
        """`rollout_batch` with the structs and every pointer resolved once: returns `launch(rng_offset)`."""
        fn = _native.load().fg_rollout_scenario
        p, d = self.params(world, auto_reset=auto_reset), self.descriptor()
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
        M = self.num_obstacles
        _native.check(_native.load().fg_reset_scenario(
            self.params(world, rng_offset), self.descriptor(), world.num_envs, len(world.agents), _native.ptr(mask),
            world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
            world.landmark_pos.data_ptr(), world.obstacle_pos.data_ptr() if M else None,
            world.obstacle_vel.data_ptr() if M else None, world.step_count.data_ptr(), _native.current_stream(world.device)))
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
        pos, _ = world.get_state()
        d = (pos - pos[:, agent.i:agent.i + 1]).norm(dim=-1)
        dl = (pos[:, :, None, :] - world.landmark_pos[:, None, :, :]).norm(dim=-1).min(1).values
        return {'reward': self.reward(agent, world),
                'collisions': (d < 2 * world.agents[0].size).sum(1),
                'min_dists': dl.sum(1), 'occupied_landmarks': (dl < 0.1).sum(1)}

    def is_collision(self, agent1, agent2):
        dist = (agent1.state.p_pos - agent2.state.p_pos).norm(dim=-1)
        return dist < (agent1.size + agent2.size)
