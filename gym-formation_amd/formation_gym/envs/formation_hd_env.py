"""formation_hd_env: Hausdorff-distance formation task (reference
formation_gym/envs/formation_hd_env.py), MI355X-native.

Same plugin surface (`class Scenario(BaseScenario)` with make_world /
reset_world / observation / reward / benchmark_data / is_collision /
generate_shape).  The arithmetic of observation (:38-59), reward (:61-75) and
the env shell's done (environment.py:172-178) runs inside the fused HIP kernel
`fg_step_hd`; the per-agent callbacks return that agent's slice.
"""
import ctypes

import numpy as np
import torch

from formation_gym import _native, placement
from formation_gym.core import World, Agent, Landmark
from formation_gym.scenario import BaseScenario


class _Plan(object):
    """Owner of a library-side launch plan (fg_step_hd_plan): keeps the tensors whose addresses the plan holds alive and
    frees the plan with the last reference to the bound launcher."""

    def __init__(self, lib, handle, keep):
        self.lib, self.handle, self.keep = lib, handle, keep

    def __del__(self):
        try:
            if self.handle:
                self.lib.fg_plan_destroy(self.handle)
                self.handle = None
        except Exception:                 # noqa: BLE001 - interpreter shutdown
            pass


class Scenario(BaseScenario):
    def make_world(self, num_agents=3, episode_length=100, num_envs=1, device=None):
        world = World(num_envs=num_envs, device=device)
        world.world_length = episode_length
        world.dim_c = 2
        world.collaborative = True
        self.num_agents = num_agents
        world.agents = [Agent() for _ in range(num_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = True
            agent.silent = True
            agent.size = 0.03
        world.landmarks = [Landmark() for _ in range(num_agents)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmarks %d' % i
            landmark.collide = False
            landmark.movable = False
            landmark.size = 0.01
        world.allocate()
        world.scenario = self
        B, N = world.num_envs, num_agents
        f = dict(dtype=torch.float32, device=world.device)
        self.ideal_shape = torch.zeros((B, N, 2), **f)     # Scenario attribute, as in :86-93
        self.ideal_vel = torch.zeros((B, 2), **f)          # :95
        self._rngs = None
        self._seed = 1
        self._cache = None
        self.reset_world(world)
        return world

    # ---- RNG: the reference draws from the global legacy MT19937 ----------
    def seed(self, seed=None):
        """env.seed(s) (environment.py:106-110).  Env b gets RandomState(s + 1000 (env_base + b)),
        the per-worker convention of train/maddpg-v2/main.py:19-30; b = 0 is the
        reference's single env.  `env_base` = global index of this batch's env 0 (sharding.make_env_shard)."""
        self._seed = 1 if seed is None else int(seed)
        self._rngs = None
        self._mt_state = None

    def _streams(self, B):
        if self._rngs is None or len(self._rngs) != B:
            self._rngs = [np.random.RandomState(self._seed + 1000 * (getattr(self, "env_base", 0) + b)) for b in range(B)]
        return self._rngs

    def reset_world(self, world, env_mask=None):
        """:77-95 - N agent positions, N landmark positions, one ideal velocity,
        each U(-1,1)^2, drawn in that order from each env's own stream; the
        ideal shape is the centred landmark set."""
        B, N = world.num_envs, len(world.agents)
        rngs = self._streams(B)
        idx = range(B) if env_mask is None else [b for b in range(B) if env_mask[b]]
        pos, vel = world.get_state()
        pos = pos.cpu().numpy().astype(np.float64); vel = vel.cpu().numpy().astype(np.float64)
        shape = self.ideal_shape.cpu().numpy().astype(np.float64)
        ivel = self.ideal_vel.cpu().numpy().astype(np.float64)
        lm = world.landmark_pos.cpu().numpy().astype(np.float64)
        for b in idx:
            rs = rngs[b]
            pos[b] = rs.uniform(-1, +1, (N, 2))
            vel[b] = 0.0
            raw = rs.uniform(-1, +1, (N, 2))
            shape[b] = raw - raw.mean(0)
            lm[b] = raw
            ivel[b] = rs.uniform(-1, +1, 2)
        world.set_state(pos, vel)
        self.ideal_shape.copy_(torch.as_tensor(shape, dtype=torch.float32))
        self.ideal_vel.copy_(torch.as_tensor(ivel, dtype=torch.float32))
        world.landmark_pos.copy_(torch.as_tensor(lm, dtype=torch.float32))
        if env_mask is None:
            world.step_count.zero_()
        else:
            m = torch.as_tensor(np.asarray(env_mask, dtype=bool), device=world.device)
            world.step_count.masked_fill_(m, 0)
        self._cache = None

    def set_formation(self, world, ideal_shape=None, ideal_vel=None):
        """Upload explicit ideal shapes [B,N,2] / velocities [B,2] (parity tests, curricula)."""
        if ideal_shape is not None:
            self.ideal_shape.copy_(torch.as_tensor(np.array(ideal_shape), dtype=torch.float32))
        if ideal_vel is not None:
            self.ideal_vel.copy_(torch.as_tensor(np.array(ideal_vel), dtype=torch.float32))
        self._cache = None

    # ---- batched protocol ----------------------------------------------------
    def obs_dim(self, world):
        return 6 * len(world.agents)

    def params(self, world, auto_reset=False, rng_offset=0, obs=None, scripted_ok=False):
        a0 = world.agents[0]
        p = world.native_params(collide_thresh=(a0.size + a0.size) / 2,   # :121
                                auto_reset=auto_reset, seed=self._seed, rng_offset=rng_offset, scripted_ok=scripted_ok)
        p.obs_env_pitch = self.obs_env_pitch(obs, len(world.agents))
        p.env_index_base = int(getattr(self, "env_base", 0))
        if world.any_non_silent():                        # :48-51 the communication block carries the others' state.c
            if world.dim_c != 2:
                raise NotImplementedError("formation_hd_env's observation has a communication block of dim_c = 2 (:40, :48-51); "
                                          "World.step / update_agent_state take any dim_c")
            p.comm_state = world.ensure_comm()[0].data_ptr()
        if obs is not None and obs.numel() and placement.is_placed(obs.data_ptr()):
            p.obs_placed = 1                              # a buffer of chunks spread over the device memory: more writer waves pay
        return p

    @staticmethod
    def obs_env_pitch(obs, N):
        """Floats between the [N, 6N] blocks of consecutive envs of an observation tensor [B, N, 6N] or
        [K, B, N, 6N]: 0 for a contiguous tensor, else its (padded) env stride.  Rows must be dense and, with a step
        axis, the step slots B env pitches apart - anything else cannot be described to the kernels."""
        if obs is None or obs.numel() == 0 or obs.is_contiguous():
            return 0
        D = 6 * N
        ok = obs.dim() in (3, 4) and obs.stride(-1) == 1 and obs.stride(-2) == D and obs.shape[-2:] == (N, D)
        B = obs.shape[-3] if ok else 0
        slots = obs.shape[0] if ok and obs.dim() == 4 else 1
        # a dimension of size 1 has no meaningful stride: the pitch is read from whichever of (env, step slot) moves
        if ok and B > 1:
            pitch = obs.stride(-3)
            if slots > 1:
                ok = obs.stride(0) == B * pitch
        elif ok and slots > 1:
            pitch = obs.stride(0)                    # one env per slot: slots are one pitch apart
        else:
            pitch = N * D
        if not ok or pitch < N * D or pitch % 2:
            raise ValueError("observation tensor must be [.., B, N, 6N] with dense rows and a uniform even env pitch "
                             ">= 6 N^2 floats; got shape %s strides %s" % (tuple(obs.shape), tuple(obs.stride())))
        if pitch == N * D:
            return 0
        return int(pitch)

    def step_batch(self, world, act, out, auto_reset=False, rng_offset=0):
        lib = _native.load()
        _native.check(lib.fg_step_hd(
            self.params(world, auto_reset, rng_offset, out.get("obs")), world.num_envs, len(world.agents),
            world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
            act.data_ptr(), self.ideal_shape.data_ptr(), self.ideal_vel.data_ptr(),
            world.step_count.data_ptr(),
            out["obs"].data_ptr(), out["reward"].data_ptr(), _native.ptr(out.get("indiv")),
            _native.ptr(out.get("done")), _native.ptr(out.get("near_lm")), _native.ptr(out.get("near_ag")),
            _native.ptr(out.get("hd_idx")), _native.current_stream(world.device)))
        self._cache = out

    def bind_step(self, world, act, out, auto_reset=False):
        """Resolve every pointer once and return `launch(rng_offset)`: the per-call
        host work is one ctypes call (rollout loops, bench.py)."""
        lib = _native.load()
        p = self.params(world, auto_reset, 0, out.get("obs"))
        args = (world.num_envs, len(world.agents),
                world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
                act.data_ptr(), self.ideal_shape.data_ptr(), self.ideal_vel.data_ptr(),
                world.step_count.data_ptr(),
                out["obs"].data_ptr(), out["reward"].data_ptr(), _native.ptr(out.get("indiv")),
                _native.ptr(out.get("done")), _native.ptr(out.get("near_lm")), _native.ptr(out.get("near_ag")),
                _native.ptr(out.get("hd_idx")), _native.current_stream(world.device))
        # the checked launch description is kept by the library (fg_step_hd_plan): a step is a two-argument call
        handle = ctypes.c_void_p()
        _native.check(lib.fg_step_hd_plan(p, *args, ctypes.byref(handle)))
        plan = _Plan(lib, handle, (act, out, p, world.pos_x, world.pos_y, world.vel_x, world.vel_y, self.ideal_shape,
                                   self.ideal_vel, world.step_count))
        fn = lib.fg_plan_launch
        raw = plan.handle

        def launch(rng_offset=0):
            rc = fn(raw, rng_offset)
            if rc:
                _native.check(rc)
            return plan
        self._cache = None
        return launch

    def observe_batch(self, world, out):
        lib = _native.load()
        _native.check(lib.fg_observe_hd(                      # (no action involved: a World with scripted agents may be observed)
            self.params(world, obs=out.get("obs"), scripted_ok=True), world.num_envs, len(world.agents),
            world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
            self.ideal_shape.data_ptr(), self.ideal_vel.data_ptr(), world.step_count.data_ptr(),
            _native.ptr(out.get("obs")), _native.ptr(out.get("reward")), _native.ptr(out.get("indiv")),
            _native.ptr(out.get("done")), _native.ptr(out.get("near_lm")), _native.ptr(out.get("near_ag")),
            _native.ptr(out.get("hd_idx")), _native.current_stream(world.device)))
        self._cache = out

    def rollout_batch(self, world, act_seq, out, obs_every=1, auto_reset=False, rng_offset=0):
        """K steps in one launch; act_seq [K,B,N,2], out tensors carry a leading K."""
        lib = _native.load()
        K = act_seq.shape[0]
        _native.check(lib.fg_rollout_hd(
            self.params(world, auto_reset, rng_offset, out.get("obs")), world.num_envs, len(world.agents), K,
            world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
            act_seq.data_ptr(), self.ideal_shape.data_ptr(), self.ideal_vel.data_ptr(),
            world.step_count.data_ptr(),
            _native.ptr(out.get("obs")), out["reward"].data_ptr(), _native.ptr(out.get("indiv")),
            _native.ptr(out.get("done")), int(obs_every), _native.current_stream(world.device)))
        self._cache = None

    def rollout_policy_batch(self, world, K, per_layer, out, obs_every=1, auto_reset=False, rng_offset=0):
        """K closed-loop steps with the built-in controller (`fg_rollout_hd_policy`): out["act"] [K,B,N,2]
        receives the actions taken, the other tensors are those of `rollout_batch`."""
        lib = _native.load()
        _native.check(lib.fg_rollout_hd_policy(
            self.params(world, auto_reset, rng_offset, out.get("obs")), world.num_envs, len(world.agents), int(K), int(per_layer),
            world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
            out["act"].data_ptr(), self.ideal_shape.data_ptr(), self.ideal_vel.data_ptr(),
            world.step_count.data_ptr(),
            _native.ptr(out.get("obs")), out["reward"].data_ptr(), _native.ptr(out.get("indiv")),
            _native.ptr(out.get("done")), int(obs_every), _native.current_stream(world.device)))
        self._cache = None

    def bind_rollout_policy(self, world, K, per_layer, out, obs_every=1, auto_reset=False):
        """`rollout_policy_batch` with every pointer and the FgParams struct resolved once: returns
        `launch(rng_offset)`, one ctypes call per closed-loop launch."""
        lib = _native.load()
        p = self.params(world, auto_reset, 0, out.get("obs"))
        args = (world.num_envs, len(world.agents), int(K), int(per_layer),
                world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
                out["act"].data_ptr(), self.ideal_shape.data_ptr(), self.ideal_vel.data_ptr(),
                world.step_count.data_ptr(),
                _native.ptr(out.get("obs")), out["reward"].data_ptr(), _native.ptr(out.get("indiv")),
                _native.ptr(out.get("done")), int(obs_every), _native.current_stream(world.device))
        fn = lib.fg_rollout_hd_policy
        keep = out

        def launch(rng_offset=0):
            p.rng_offset = rng_offset
            rc = fn(p, *args)
            if rc:
                _native.check(rc)
            return keep
        return launch

    def policy_actions(self, world, per_layer, out=None):
        """get_action_BFS(ezpolicy, obs, per_layer) for the CURRENT state of every env, straight from the
        simulator state (`fg_policy_bfs_state`): raw actions [B, N, 2]."""
        B, N = world.num_envs, len(world.agents)
        if out is None:
            out = torch.empty((B, N, 2), dtype=torch.float32, device=world.device)
        _native.check(_native.load().fg_policy_bfs_state(
            B, N, int(per_layer), world.pos_x.data_ptr(), world.pos_y.data_ptr(),
            self.ideal_shape.data_ptr(), self.ideal_vel.data_ptr(), out.data_ptr(),
            _native.current_stream(world.device)))
        return out

    def bind_rollout(self, world, act_seq, out, obs_every=1, auto_reset=False):
        """`rollout_batch` with every pointer and the FgParams struct resolved once: returns
        `launch(rng_offset)`, one ctypes call per K-step launch (a 9 x 4096 launch lasts ~35 us on the GPU,
        less than the generic path spends in Python)."""
        lib = _native.load()
        p = self.params(world, auto_reset, 0, out.get("obs"))
        args = (world.num_envs, len(world.agents), int(act_seq.shape[0]),
                world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
                act_seq.data_ptr(), self.ideal_shape.data_ptr(), self.ideal_vel.data_ptr(),
                world.step_count.data_ptr(),
                _native.ptr(out.get("obs")), out["reward"].data_ptr(), _native.ptr(out.get("indiv")),
                _native.ptr(out.get("done")), int(obs_every), _native.current_stream(world.device))
        fn = lib.fg_rollout_hd
        keep = (act_seq, out)

        def launch(rng_offset=0):
            p.rng_offset = rng_offset
            rc = fn(p, *args)
            if rc:
                _native.check(rc)
            return keep
        return launch

    def upload_mt_streams(self, world):
        """Copy every env's legacy MT19937 state (RandomState(seed + 1000 b), at its CURRENT
        position) to the device; from here on `reset_mt` continues those streams on the GPU."""
        B = world.num_envs
        st = np.zeros((B, 626), dtype=np.uint32)
        for b, rs in enumerate(self._streams(B)):
            _, key, pos = rs.get_state()[:3]
            st[b, :624] = key
            st[b, 624] = pos
        self._mt_state = torch.as_tensor(st.view(np.int32)).to(world.device)
        return self._mt_state

    def reset_mt(self, world, mask=None):
        """Scenario.reset_world on the GPU from the reference's own RNG streams (bit-exact with
        the host path `reset_world`): masked envs draw N agent positions, N landmark positions
        and the ideal velocity from their MT19937 state, which advances in place."""
        if getattr(self, "_mt_state", None) is None or self._mt_state.shape[0] != world.num_envs:
            self.upload_mt_streams(world)
        lib = _native.load()
        _native.check(lib.fg_reset_hd_mt(
            world.num_envs, len(world.agents), _native.ptr(mask), self._mt_state.data_ptr(),
            world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
            self.ideal_shape.data_ptr(), self.ideal_vel.data_ptr(), world.landmark_pos.data_ptr(),
            world.step_count.data_ptr(), _native.current_stream(world.device)))
        self._cache = None

    def reset_mt_done(self, world, obs=None):
        """The vec-env worker's `if all(done): ob = env.reset()` (env_wrappers.py:14-18) for every env at once, decided on
        the device (`fg_reset_hd_mt_done`): envs whose step counter has reached world_length restart from their own
        MT19937 streams and, with `obs` [B, N, 6N], get their reset observation written over the step's."""
        if getattr(self, "_mt_state", None) is None or self._mt_state.shape[0] != world.num_envs:
            self.upload_mt_streams(world)
        _native.check(_native.load().fg_reset_hd_mt_done(
            world.num_envs, len(world.agents), int(world.world_length), self._mt_state.data_ptr(),
            world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
            self.ideal_shape.data_ptr(), self.ideal_vel.data_ptr(), world.landmark_pos.data_ptr(),
            world.step_count.data_ptr(), _native.ptr(obs), self.obs_env_pitch(obs, len(world.agents)),
            _native.current_stream(world.device)))
        self._cache = None

    def bind_reset_mt_done(self, world, obs=None):
        """`reset_mt_done` with every pointer resolved once: returns `launch()`, one ctypes call per vec-env step."""
        if getattr(self, "_mt_state", None) is None or self._mt_state.shape[0] != world.num_envs:
            self.upload_mt_streams(world)
        fn = _native.load().fg_reset_hd_mt_done
        args = (world.num_envs, len(world.agents), int(world.world_length), self._mt_state.data_ptr(),
                world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
                self.ideal_shape.data_ptr(), self.ideal_vel.data_ptr(), world.landmark_pos.data_ptr(),
                world.step_count.data_ptr(), _native.ptr(obs), self.obs_env_pitch(obs, len(world.agents)),
                _native.current_stream(world.device))
        keep = (self._mt_state, obs)

        def launch():
            rc = fn(*args)
            if rc:
                _native.check(rc)
            self._cache = None
            return keep
        return launch

    def reset_device(self, world, mask=None, rng_offset=0):
        """Throughput-mode reset on the GPU (counter RNG, distributional parity only)."""
        lib = _native.load()
        _native.check(lib.fg_reset_hd(
            self.params(world, rng_offset=rng_offset), world.num_envs, len(world.agents), _native.ptr(mask),
            world.pos_x.data_ptr(), world.pos_y.data_ptr(), world.vel_x.data_ptr(), world.vel_y.data_ptr(),
            self.ideal_shape.data_ptr(), self.ideal_vel.data_ptr(), world.step_count.data_ptr(),
            _native.current_stream(world.device)))
        self._cache = None

    # ---- per-agent callbacks (reference signature) -----------------------------
    def _fresh(self, world):
        if self._cache is None:
            B, N = world.num_envs, len(world.agents)
            f = dict(dtype=torch.float32, device=world.device)
            out = dict(obs=torch.empty((B, N, 6 * N), **f), reward=torch.empty((B, N), **f),
                       indiv=torch.empty((B, N), **f))
            self.observe_batch(world, out)
        return self._cache

    def observation(self, agent, world):
        """:38-59 for `agent` in every env -> [B, 6N].  Also applies the
        reference's side effect of re-centring the landmarks on the agents (:40-44)."""
        out = self._fresh(world)
        cen = torch.stack((world.pos_x.mean(1), world.pos_y.mean(1)), -1) - world.landmark_pos.mean(1)
        world.landmark_pos += cen[:, None, :]
        return out["obs"][:, agent.i]

    def reward(self, agent, world):
        """:61-75 individual reward of `agent` in every env -> [B]."""
        return self._fresh(world)["indiv"][:, agent.i]

    def benchmark_data(self, agent, world):
        """:97-117 for `agent` in every env -> dict of [B] tensors.  The env shell evaluates it right after
        `observation` (environment.py:127-131), which has just re-centred the landmarks on the agents (:40-44):
        the landmark positions it measures against are ideal_shape + centroid(agents)."""
        rew = self.reward(agent, world)
        pos, _ = world.get_state()
        d = (pos - pos[:, agent.i:agent.i + 1]).norm(dim=-1)
        a0 = world.agents[0]
        collisions = (d < (a0.size + a0.size) / 2).sum(1)       # is_collision (:119-121), `agent` itself included (:101-104)
        lm = self.ideal_shape + pos.mean(1, keepdim=True)
        dl = (pos[:, :, None, :] - lm[:, None, :, :]).norm(dim=-1).min(1).values     # per landmark: its nearest agent
        return {'reward': rew, 'collisions': collisions, 'min_dists': dl.sum(1),
                'occupied_landmarks': (dl < 0.1).sum(1)}

    def is_collision(self, agent1, agent2):
        """:119-121 -> bool [B]."""
        dist = (agent1.state.p_pos - agent2.state.p_pos).norm(dim=-1)
        return dist < (agent1.size + agent2.size) / 2

    def generate_shape(self, layer, layer_shapes=None):
        """:123-139 default hierarchical target shape, [3, ..., 3, 2] nested like the reference."""
        table = np.array([
            [[0, -1], [0.5, 0], [0, 1]],
            [[0, 1.6], [-1, 0], [1, 0]],
            [[1.5, 0], [0, 0], [-1.5, 0]],
            [[0, 0.6], [1, 0], [-1, 0]],
        ]) if layer_shapes is None else np.asarray(layer_shapes)
        assert layer < table.shape[0], 'Layer shape is not enough!'
        if layer == 0:
            return table[0]
        inner = self.generate_shape(layer - 1, layer_shapes)
        return np.array([table[layer][i] + inner * 0.45 for i in range(table.shape[1])])
