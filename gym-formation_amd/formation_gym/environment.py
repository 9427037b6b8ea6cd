"""MultiAgentEnv: the env shell of reference formation_gym/environment.py
(:11-236), driving B environments per call on one MI355X.

Drop-in surface kept: constructor signature, `seed`, `reset`, `step`,
`action_space[]`, `observation_space[]`, `share_observation_space[]`,
`num_agents`, `agents`, `world`, `world_length`, `current_step`, `num_envs`,
`shared_reward`.

Two calling conventions for `step`:
  * reference style - `action_n` is a list of N arrays of shape (2,) (num_envs
    must be 1): returns `(obs_n, reward_n, done_n, info_n)` exactly shaped like
    the reference (lists of float64 arrays / [float] / bool / dict) and, like the
    reference (environment.py:216-221), scales the caller's arrays in place and
    rejects plain Python lists;
  * batched - `action_n` is a float32 tensor [B, N, 2] on the device: returns
    device tensors obs [B,N,6N], reward [B,N,1], done [B,N] (bool) and
    info = {'individual_reward': [B,N]}.  These are views of buffers that the
    next `step` overwrites.
Either way ONE fused HIP launch does `_set_action` (x N), `world.step()` and the
N observation / 2N reward / N done callbacks of environment.py:113-142.
"""
import numpy as np
import torch

from . import _native, spaces

cam_range = 2


class MultiAgentEnv(object):
    metadata = {'render.modes': ['human', 'rgb_array']}

    def __init__(self, world, reset_callback=None, reward_callback=None,
                 observation_callback=None, info_callback=None,
                 done_callback=None, post_step_callback=None,
                 shared_viewer=True, discrete_action=False):
        self.world = world
        self.world_length = self.world.world_length
        self.current_step = 0
        self.agents = self.world.policy_agents
        self.num_agents = len(world.policy_agents)
        self.reset_callback = reset_callback
        self.reward_callback = reward_callback
        self.observation_callback = observation_callback
        self.info_callback = info_callback
        self.done_callback = done_callback
        self.post_step_callback = post_step_callback
        self.num_envs = world.num_envs
        self.scenario = world.scenario
        if self.scenario is None:
            raise ValueError("world.scenario is not set: make_world() must attach the Scenario "
                             "that owns the batched kernels")
        # action modes (environment.py:39-47): make_env never sets them, direct construction may
        self.discrete_action_space = bool(discrete_action)        # 5-vector per agent, u = (a1-a2, a3-a4)
        self.discrete_action_input = False                        # an index 0..4 per agent
        self.force_discrete_action = bool(getattr(world, 'discrete_action', False))   # arg-max one-hot
        self.shared_reward = world.collaborative if hasattr(world, 'collaborative') else False
        self.time = 0
        self.auto_reset = False           # vec-env worker semantics, see vec_env.py
        self._rng_offset = 0

        # spaces (environment.py:55-96)
        self.action_space = []
        self.observation_space = []
        obs_dim = self.scenario.obs_dim(world)
        share_obs_dim = 0
        for agent in self.agents:
            if self.discrete_action_space:                        # :64-65
                u_space = spaces.Discrete(world.dim_p * 2 + 1)
            else:
                u_space = spaces.Box(low=-agent.u_range, high=+agent.u_range, shape=(world.dim_p,), dtype=np.float32)
            if not agent.silent:                                  # :72-84: a Tuple (physical, communication) action space
                c_space = spaces.Discrete(world.dim_c) if self.discrete_action_space else \
                    spaces.Box(low=0.0, high=1.0, shape=(world.dim_c,), dtype=np.float32)
                self.action_space.append(spaces.Tuple([u_space, c_space]))
            else:
                self.action_space.append(u_space)
            share_obs_dim += obs_dim
            self.observation_space.append(spaces.Box(low=-np.inf, high=+np.inf,
                                                     shape=(obs_dim,), dtype=np.float32))
        self.share_observation_space = [spaces.Box(low=-np.inf, high=+np.inf, shape=(share_obs_dim,),
                                                   dtype=np.float32) for _ in range(self.num_agents)]

        B, N = self.num_envs, self.num_agents
        dev = world.device
        f = dict(dtype=torch.float32, device=dev)
        # the four per-step outputs are views of ONE allocation, so that the reference-style API
        # (num_envs == 1: NumPy lists in and out) brings a whole step back in a single device-to-host copy
        n_obs, n_bn = B * N * obs_dim, B * N
        self._flat = torch.zeros(n_obs + 2 * n_bn + (n_bn + 3) // 4, **f)
        self._out = dict(
            obs=self._flat[:n_obs].view(B, N, obs_dim),
            reward=self._flat[n_obs:n_obs + n_bn].view(B, N),
            indiv=self._flat[n_obs + n_bn:n_obs + 2 * n_bn].view(B, N),
            done=self._flat[n_obs + 2 * n_bn:].view(torch.uint8)[:n_bn].view(B, N),
        )
        self._act = torch.zeros((B, N, 2), **f)
        self._host = self._act_host = None
        if B == 1 and torch.device(dev).type == "cuda":       # pinned mirrors for the single-env list API
            self._host = torch.empty(self._flat.shape, dtype=torch.float32).pin_memory()
            self._act_host = torch.zeros((1, N, 2), dtype=torch.float32).pin_memory()
        self._launchers = {}              # pre-bound step launches, see _bound_step
        self._roll_launchers = {}         # pre-bound K-step launches into caller-owned buffers, see rollout
        self.placement = None             # report of the last buffer placement probe (alloc_rollout_buffers)
        self._auto_out = {}               # (K, obs_every, policy) -> placed output buffers of rollout(out=None), see _default_out
        # rollout(out=None): env-owned, placed, re-used buffers (False: fresh tensors per call)
        self.default_placed = bool(getattr(self.scenario, "DEFAULT_PLACED", True))
        self.shared_viewer = shared_viewer
        self.viewers = [None]

    # ------------------------------------------------------------------ seed
    def seed(self, seed=None):
        """environment.py:106-110 (default seed 1)."""
        self.scenario.seed(seed)

    def enable_assignments(self, on=True):
        """Also emit the landmark-index assignments (nearest ideal point per
        agent, nearest agent per ideal point, Hausdorff witness pairs) each step."""
        self._launchers.clear()           # bound launches hold the old output pointers
        B, N = self.num_envs, self.num_agents
        dev = self.world.device
        if on:
            self._out["near_lm"] = torch.zeros((B, N), dtype=torch.int32, device=dev)
            self._out["near_ag"] = torch.zeros((B, N), dtype=torch.int32, device=dev)
            self._out["hd_idx"] = torch.zeros((B, 4), dtype=torch.int32, device=dev)
        else:
            for k in ("near_lm", "near_ag", "hd_idx"):
                self._out.pop(k, None)

    # ------------------------------------------------------------------ step
    def step(self, action_n):
        self.current_step += 1
        self.agents = self.world.policy_agents
        if self.world.any_frozen_silent():
            # environment.py:191-236: `_set_action` consumes the action only `if agent.movable` (or as the communication of
            # a non-silent agent) and then asserts that nothing is left - a silent immovable agent trips that assertion
            # (fixture hd_n6_immovable records it).  Such agents are driven through the World API: world.step().
            raise AssertionError
        if self.world.any_non_silent():
            # The reference cannot step non-silent (movable) agents either: `_set_action` consumes the whole action for
            # the physical part and then indexes the exhausted list for the communication part (environment.py:216-231;
            # fixture hd_n5_comm records the IndexError).  Non-silent agents are driven through the World API instead:
            # agent.action.u / agent.action.c, world.step(), scenario.observation / reward (core.py:206-225, 279-286).
            raise IndexError("list index out of range")
        batched = torch.is_tensor(action_n)
        mode = self._action_mode()
        if mode:
            act = self._decode_actions(action_n, mode, batched)
        elif batched:
            act = action_n
            if act.shape != self._act.shape:
                raise ValueError("batched action must have shape %s, got %s"
                                 % (tuple(self._act.shape), tuple(act.shape)))
            if act.dtype != torch.float32 or act.device != self._act.device or not act.is_contiguous():
                self._act.copy_(act)
                act = self._act
        else:
            act = self._stage_reference_actions(action_n)
        off = self._launch_rng_offset()
        if not self._bound_step(act, off):
            self.scenario.step_batch(self.world, act, self._out, auto_reset=self.auto_reset, rng_offset=off)
        self._advance_rng(1)
        self.world.world_step += 1
        if self.post_step_callback is not None:
            self.post_step_callback(self.world)
        if batched:
            return self._batched_result()
        return self._reference_result()

    def rollout(self, action_seq, out=None, obs_every=1):
        """K consecutive `step` calls in ONE launch (`fg_rollout_hd`): for policies that hand over a
        whole action sequence (open-loop / pre-staged random policies, model-predictive candidates).
        `action_seq` is a [K, B, N, 2] tensor of raw continuous actions.  Results are bit-identical to
        K calls of `step` (device auto-reset included, when `self.auto_reset` is set) and come back
        stacked along a leading step axis:
            obs [K // obs_every, B, N, D]  (every obs_every-th step; obs_every = 1: every step)
            reward [K, B, N, 1], done [K, B, N] bool, info {'individual_reward': [K, B, N]}
        `out` may hold pre-allocated buffers (keys obs, reward, indiv, done with those shapes, done as
        uint8), so that a training loop re-uses them.  out=None (the default): the env's own buffers for this (K,
        obs_every), made on first use with the observation buffer PLACED (`alloc_rollout_buffers`: one probe per shape,
        0.2-0.4 s and transiently up to 6 x the buffer of device memory - the implicit path never escalates to larger
        arenas) and re-used by every later call.  LIKE `step`, THE RESULTS ARE VIEWS OF ENV-OWNED BUFFERS THAT THE NEXT
        `rollout` OF THE SAME SHAPE OVERWRITES: `traj.append(env.rollout(a))` aliases - clone what must outlive the next
        call, or pass out=False (fresh, ordinary tensors on every call) or your own `out`.  The four most recently used
        (K, obs_every, policy) shapes keep their buffers; `env.default_placed = False` makes out=None mean out=False."""
        roll = getattr(self.scenario, "rollout_batch", None)
        if roll is None or self.post_step_callback is not None:
            # no multi-step launch (a reference-style Scenario file: its callbacks run on the host), or a post_step_callback
            # (environment.py:140-141: a host function after every step): K `step` calls behind the same interface
            return self._rollout_by_steps(action_seq, out, int(obs_every))
        mode = self._action_mode()
        if mode:
            # the discrete action modes of _set_action (environment.py:194-215): the whole sequence is decoded to raw u in
            # ONE launch (`fg_decode_actions` over K B N entries), then the K steps run as for continuous actions
            act = self._decode_action_seq(action_seq, mode)
            K = int(act.shape[0])
        else:
            if not torch.is_tensor(action_seq) or action_seq.dim() != 4 or tuple(action_seq.shape[1:]) != tuple(self._act.shape):
                raise ValueError("action_seq must be a tensor of shape [K, %d, %d, 2]" % tuple(self._act.shape[:2]))
            K = int(action_seq.shape[0])
            act = action_seq
            if act.dtype != torch.float32 or act.device != self._act.device or not act.is_contiguous():
                act = act.to(device=self._act.device, dtype=torch.float32).contiguous()
        obs_every = int(obs_every)
        if K < 1 or obs_every < 1:
            raise ValueError("need K >= 1 steps and obs_every >= 1")
        B, N = self.num_envs, self.num_agents
        D = self._out["obs"].shape[-1]
        if out is None:
            out = self._default_out(K, obs_every, False) if self.default_placed else False
        own_buffers = out is not False
        if out is False:
            f = dict(dtype=torch.float32, device=self._act.device)
            out = dict(obs=torch.empty((K // obs_every, B, N, D), **f), reward=torch.empty((K, B, N), **f),
                       indiv=torch.empty((K, B, N), **f),
                       done=torch.zeros((K, B, N), dtype=torch.uint8, device=self._act.device))
        # launches into CALLER-OWNED buffers are bound once per (actions, buffers, stream, constants): a training
        # loop that re-uses its buffers pays one ctypes call per launch (cf. _bound_step).  A binding keeps its
        # tensors alive (their addresses are baked in), so the cache is small and skips internally allocated outputs.
        bind = getattr(self.scenario, "bind_rollout", None)
        key = None
        if own_buffers and bind is not None and all(k in out for k in ("obs", "reward", "indiv", "done")):
            key = ("roll", act.data_ptr(), K, out["obs"].data_ptr(), tuple(out["obs"].stride()), out["reward"].data_ptr(),
                   out["indiv"].data_ptr(), out["done"].data_ptr(), obs_every, self.auto_reset,
                   _native.current_stream_fast(self.world.device),
                   self.world.params_signature(), getattr(self.scenario, "_seed", 0))
        launch = self._roll_launchers.get(key) if key is not None else None
        if launch is None:
            want = dict(obs=(K // obs_every, B, N, D), reward=(K, B, N), indiv=(K, B, N), done=(K, B, N))
            for k, shp in want.items():
                if k not in out or tuple(out[k].shape) != shp or not (out[k].is_contiguous() or k == "obs"):
                    raise ValueError("out[%r] must be a contiguous tensor of shape %s" % (k, shp))   # obs: or a padded env pitch
            if key is not None:
                if len(self._roll_launchers) >= 8:
                    self._roll_launchers.clear()
                launch = self._roll_launchers[key] = bind(self.world, act, out, obs_every=obs_every,
                                                          auto_reset=self.auto_reset)
        if launch is not None:
            launch(self._launch_rng_offset())
            self.scenario._cache = None
        else:
            roll(self.world, act, out, obs_every=obs_every, auto_reset=self.auto_reset, rng_offset=self._launch_rng_offset())
        self._advance_rng(K)
        self.current_step += K
        self.world.world_step += K
        rew = out["reward"] if self.shared_reward else out["indiv"]
        return out["obs"], rew.unsqueeze(-1), out["done"].view(torch.bool), {"individual_reward": out["indiv"]}

    def _rollout_by_steps(self, action_seq, out, obs_every):
        """`rollout` as K calls of `step` (host-paced: one launch - and whatever the scenario / the post_step_callback does on
        the host - per step), results stacked like `rollout`'s."""
        K = int(len(action_seq))
        if K < 1 or obs_every < 1:
            raise ValueError("need K >= 1 steps and obs_every >= 1")
        B, N = self.num_envs, self.num_agents
        D = self._out["obs"].shape[-1]
        dev = self._act.device
        if out is None or out is False:
            f = dict(dtype=torch.float32, device=dev)
            out = dict(obs=torch.empty((K // obs_every, B, N, D), **f), reward=torch.empty((K, B, N), **f),
                       indiv=torch.empty((K, B, N), **f), done=torch.zeros((K, B, N), dtype=torch.uint8, device=dev))
        want = dict(obs=(K // obs_every, B, N, D), reward=(K, B, N), indiv=(K, B, N), done=(K, B, N))
        for k, shp in want.items():
            if k not in out or tuple(out[k].shape) != shp:
                raise ValueError("out[%r] must be a tensor of shape %s" % (k, shp))
        first = None if self.shared_reward else torch.empty((K, B, N), dtype=torch.float32, device=dev)
        for k in range(K):
            act = action_seq[k]
            if torch.is_tensor(act) and (act.dtype != torch.float32 or act.device != dev) and not self._action_mode():
                act = act.to(device=dev, dtype=torch.float32)
            o, r, d, info = self.step(act if torch.is_tensor(act) else torch.as_tensor(act, device=dev))
            out["reward"][k].copy_(self._out["reward"]); out["indiv"][k].copy_(self._out["indiv"])
            out["done"][k].copy_(self._out["done"])
            if first is not None:
                first[k].copy_(r[..., 0])
            if (k + 1) % obs_every == 0:
                out["obs"][k // obs_every].copy_(o)
        rew = out["reward"] if self.shared_reward else first
        return out["obs"], rew.unsqueeze(-1), out["done"].view(torch.bool), {"individual_reward": out["indiv"]}

    def rollout_policy(self, K, num_agents_per_layer=3, out=None, obs_every=1):
        """The reference's demo loop (test.py:17-27) for K steps in one call:
            act_n = get_action_BFS(ezpolicy, obs_n, num_agents_per_layer); obs_n, ... = env.step(act_n)
        starting from the current state (`fg_rollout_hd_policy`; ONE launch at 3, 9, 27, 81 or 243 agents with
        3 agents per layer).  Results equal K x (`get_action_BFS` on the last observation, `step`) bit for bit and
        come back like `rollout`'s, with the actions taken under info['actions'] [K, B, N, 2]."""
        roll = getattr(self.scenario, "rollout_policy_batch", None)
        if roll is None:
            raise NotImplementedError("%s has no built-in controller" % type(self.scenario).__name__)
        if self._action_mode():
            raise NotImplementedError("the built-in controller emits raw continuous actions")
        K, obs_every = int(K), int(obs_every)
        if self.post_step_callback is not None:
            # a host function after every step (environment.py:140-141): the demo loop itself, launch by launch
            from .policy_bfs import ezpolicy, get_action_BFS
            if K < 1 or obs_every < 1:
                raise ValueError("need K >= 1 steps and obs_every >= 1")
            acts = torch.empty((K,) + tuple(self._act.shape), dtype=torch.float32, device=self._act.device)
            res = {k: [] for k in ("obs", "rew", "done", "indiv")}
            obs = self._out["obs"]
            for k in range(K):
                acts[k].copy_(get_action_BFS(ezpolicy, obs, int(num_agents_per_layer)))
                obs, r, d, info = self.step(acts[k])
                if (k + 1) % obs_every == 0:
                    res["obs"].append(obs.clone())
                res["rew"].append(r.clone()); res["done"].append(d.clone()); res["indiv"].append(info["individual_reward"].clone())
            return (torch.stack(res["obs"]) if res["obs"] else obs.new_empty((0,) + tuple(obs.shape)), torch.stack(res["rew"]),
                    torch.stack(res["done"]), {"individual_reward": torch.stack(res["indiv"]), "actions": acts})
        if K < 1 or obs_every < 1:
            raise ValueError("need K >= 1 steps and obs_every >= 1")
        B, N = self.num_envs, self.num_agents
        D = self._out["obs"].shape[-1]
        f = dict(dtype=torch.float32, device=self._act.device)
        want = dict(obs=(K // obs_every, B, N, D), reward=(K, B, N), indiv=(K, B, N), done=(K, B, N), act=(K, B, N, 2))
        if out is None:
            out = self._default_out(K, obs_every, True) if self.default_placed else False
        own_buffers = out is not False
        if out is False:
            out = {k: (torch.zeros(shp, dtype=torch.uint8, device=self._act.device) if k == "done"
                       else torch.empty(shp, **f)) for k, shp in want.items()}
        # launches into CALLER-OWNED buffers are bound once (cf. rollout): a loop that steps K = 1 at a time pays one
        # ctypes call per launch
        bind = getattr(self.scenario, "bind_rollout_policy", None)
        key = None
        if own_buffers and bind is not None and all(k in out for k in want):
            key = ("pol", K, int(num_agents_per_layer), tuple(out[k].data_ptr() for k in sorted(want)),
                   tuple(out["obs"].stride()), obs_every,
                   self.auto_reset, _native.current_stream_fast(self.world.device), self.world.params_signature(),
                   getattr(self.scenario, "_seed", 0))
        launch = self._roll_launchers.get(key) if key is not None else None
        if launch is None:
            for k, shp in want.items():
                if k not in out or tuple(out[k].shape) != shp or not (out[k].is_contiguous() or k == "obs"):
                    raise ValueError("out[%r] must be a contiguous tensor of shape %s" % (k, shp))   # obs: or a padded env pitch
            if key is not None:
                if len(self._roll_launchers) >= 8:
                    self._roll_launchers.clear()
                launch = self._roll_launchers[key] = bind(self.world, K, num_agents_per_layer, out, obs_every=obs_every,
                                                          auto_reset=self.auto_reset)
        if launch is not None:
            launch(self._launch_rng_offset())
            self.scenario._cache = None
        else:
            roll(self.world, K, num_agents_per_layer, out, obs_every=obs_every, auto_reset=self.auto_reset,
                 rng_offset=self._launch_rng_offset())
        self._advance_rng(K)
        self.current_step += K
        self.world.world_step += K
        rew = out["reward"] if self.shared_reward else out["indiv"]
        return out["obs"], rew.unsqueeze(-1), out["done"].view(torch.bool), \
            {"individual_reward": out["indiv"], "actions": out["act"]}

    # ------------------------------------------------------------ buffers
    def _default_out(self, K, obs_every, policy):
        """The env's own output buffers of `rollout(out=None)` / `rollout_policy(out=None)` for this shape: placed on
        first use, kept for the next call (four shapes at most: a fifth evicts the least recently used, whose arena goes
        back to the driver with its last tensor).  The implicit probe looks at ONE arena (6 x the buffer): it never takes
        the escalation stages a caller of `alloc_rollout_buffers` may ask for (up to half of the free memory, seconds)."""
        key = (int(K), int(obs_every), bool(policy))
        out = self._auto_out.pop(key, None)
        if out is None:
            if getattr(self, "_placing", False):       # the probe's own timing launches bring their buffers
                raise RuntimeError("rollout(out=None) inside a placement probe")
            while len(self._auto_out) >= 4:
                self._auto_out.pop(next(iter(self._auto_out)))
            self._roll_launchers.clear()               # bindings keep evicted buffers alive
            out = self.alloc_rollout_buffers(K, obs_every=obs_every, policy=policy, escalate=False)
        self._auto_out[key] = out                      # most recently used last
        return out

    def _snapshot(self):
        """Everything a launch mutates (device state + host counters), for probes that must leave the env untouched."""
        w, sc = self.world, self.scenario
        dev = {k: getattr(w, k).clone() for k in ("pos_x", "pos_y", "vel_x", "vel_y", "step_count", "landmark_pos",
                                                  "obstacle_pos", "obstacle_vel")     # obstacles move; resets re-draw landmarks
               if torch.is_tensor(getattr(w, k, None))}
        scn = {k: getattr(sc, k).clone() for k in ("ideal_shape", "ideal_vel") if torch.is_tensor(getattr(sc, k, None))}
        ctr = None if w.rng_counter is None else w.rng_counter.clone()
        own = sc.snapshot_state() if hasattr(sc, "snapshot_state") else None      # a tensor scenario's per-env tensors, its generator
        return dev, scn, ctr, (self._rng_offset, self.current_step, w.world_step), own

    def _restore(self, snap):
        dev, scn, ctr, host, own = snap
        w, sc = self.world, self.scenario
        if own is not None:
            sc.restore_state(own)
        for k, v in dev.items():
            getattr(w, k).copy_(v)
        for k, v in scn.items():
            getattr(sc, k).copy_(v)
        if ctr is not None:
            w.rng_counter.copy_(ctr)
        self._rng_offset, self.current_step, w.world_step = host
        sc._cache = None

    def alloc_rollout_buffers(self, K, obs_every=1, obs_env_pitch=0, policy=False, candidates=8, mem_fraction=0.5,
                              max_arena_bytes=None, escalate=True):
        """Output buffers for `rollout` / `rollout_policy` launches of K steps, with the observation buffer - 99 % of
        the bytes - PLACED: when it is larger than the Infinity Cache, candidate buffers are composed of the chunks of a
        small arena of device memory (6 x the buffer up to 48 GiB, at least 1.5 x the buffer, never more than
        `mem_fraction` of the free memory; `max_arena_bytes` overrides), this env's own K-step launch is timed on
        `candidates` or more of them and the fastest is kept, every other chunk released (formation_gym/placement.py: the
        rate of a launch depends on which physical memory its buffer is composed of, by 10-20 %; ~0.2 s); where the arena
        cannot be made, on up to `candidates` whole allocations.  candidates < 2 switches the probe off.  The env's state
        is restored afterwards, also when the probe fails.  escalate=False: one arena only (an arena whose winner gains
        nothing is otherwise followed by up to two four times larger ones).  `obs_env_pitch` (floats, 0 = contiguous) asks for padded env blocks.  The probe's report
        is left in `self.placement`.  Returns the `out` dict to pass to `rollout(..., out=out)`.  The arena lives exactly
        as long as the observation tensor (or any view of it): dropping the tensors gives the memory back."""
        from . import placement
        K, obs_every = int(K), int(obs_every)
        B, N = self.num_envs, self.num_agents
        D = self._out["obs"].shape[-1]
        dev = self._act.device
        f = dict(dtype=torch.float32, device=dev)
        slots = K // obs_every
        pitch = int(obs_env_pitch) if obs_env_pitch else N * D
        if pitch < N * D or pitch % 2:
            raise ValueError("obs_env_pitch must be 0 or an even number of floats >= %d" % (N * D))
        small = dict(reward=torch.empty((K, B, N), **f), indiv=torch.empty((K, B, N), **f),
                     done=torch.zeros((K, B, N), dtype=torch.uint8, device=dev))
        if policy:
            small["act"] = torch.empty((K, B, N, 2), **f)

        def shaped(flat):
            return flat.view(slots, B, pitch)[:, :, :N * D].view(slots, B, N, D)

        def alloc():
            return shaped(torch.empty(slots * B * pitch, **f))

        nbytes = slots * B * pitch * 4
        if slots == 0 or nbytes < placement.MIN_PROBE_BYTES or candidates < 2:
            self.placement = {"tried": 1, "probed": False}
            return dict(small, obs=alloc())
        snap = self._snapshot()
        acts = None if policy else torch.zeros((K, B, N, 2), **f)

        per_layer = self._policy_per_layer() if policy else 0

        def time_fn(obs):
            out = dict(small, obs=obs)
            if policy:
                self.rollout_policy(K, per_layer, out=out, obs_every=obs_every)
            else:
                self.rollout(acts, out=out, obs_every=obs_every)
            self._roll_launchers.clear()               # one binding per candidate window: do not let them pile up

        arena = None
        self._placing = True
        try:
            placed = placement.probe_arena(slots * B * pitch, lambda flat: time_fn(shaped(flat)), dev, trials=candidates,
                                           mem_fraction=mem_fraction, max_arena_bytes=max_arena_bytes, escalate=escalate)
            if placed is not None:
                flat, report, arena = placed
                obs = shaped(flat)
            else:
                obs, report = placement.probe_allocation(alloc, time_fn, nbytes, dev, candidates=candidates, mem_fraction=mem_fraction)
        finally:
            # whatever happened inside the probe (out of memory, a failed launch): the env is usable and unchanged afterwards
            self._placing = False
            self._roll_launchers.clear()               # bindings made on the candidates keep them alive: drop them,
            torch.cuda.empty_cache()                   # then hand the losers back to the driver
            self._restore(snap)
        report["buffer_MB"] = round(nbytes / 1e6, 1)
        if report.get("probed") and D == 6 * N:
            alg = _native.step_hd_bytes(N) * B * K
            report["kept_GBps"] = round(alg / (report["kept_ms"] * 1e-3) / 1e9, 1)
            report["worst_GBps"] = round(alg / (report["worst_ms"] * 1e-3) / 1e9, 1)
        self.placement = report
        return dict(small, obs=obs)

    def _policy_per_layer(self):
        """Agents per layer of the built-in controller's hierarchy for this agent count (N = per^L): 3 where it fits (the
        reference's README.md:34-36), else the smallest of 2 ... 8 that does."""
        N = self.num_agents
        for per in (3, 2, 4, 5, 6, 7, 8):
            n = per
            while n < N:
                n *= per
            if n == N:
                return per
        raise ValueError("the built-in controller needs N = per^L agents with 2 <= per <= 8, got %d" % N)

    def place_step_buffers(self, candidates=8, mem_fraction=0.5):
        """The same placement for the per-step output buffer `step` writes into (only batches whose single-step
        observation tensor exceeds the Infinity Cache: 243 agents x >= 200 envs, 81 x >= 1700, 27 x >= 15 000)."""
        from . import placement
        B, N = self.num_envs, self.num_agents
        obs_dim = self._out["obs"].shape[-1]
        n_obs, n_bn = B * N * obs_dim, B * N
        nflat = n_obs + 2 * n_bn + (n_bn + 3) // 4
        dev = self._act.device
        if n_obs * 4 < placement.MIN_PROBE_BYTES or candidates < 2:
            self.placement = {"tried": 1, "probed": False}
            return self.placement
        extra = {k: v for k, v in self._out.items() if k not in ("obs", "reward", "indiv", "done")}

        def views(flat):
            return dict(extra, obs=flat[:n_obs].view(B, N, obs_dim), reward=flat[n_obs:n_obs + n_bn].view(B, N),
                        indiv=flat[n_obs + n_bn:n_obs + 2 * n_bn].view(B, N),
                        done=flat[n_obs + 2 * n_bn:].view(torch.uint8)[:n_bn].view(B, N))

        snap = self._snapshot()
        old_flat, self._flat, self._out = self._flat, None, None
        del old_flat
        torch.cuda.empty_cache()
        act = torch.zeros_like(self._act)

        def alloc():
            return torch.zeros(nflat, dtype=torch.float32, device=dev)

        def time_fn(flat):
            self.scenario.step_batch(self.world, act, views(flat), auto_reset=self.auto_reset, rng_offset=1)

        placed = placement.probe_arena(nflat, time_fn, dev, trials=candidates, mem_fraction=mem_fraction)
        if placed is not None:
            flat, report, arena = placed
            flat.zero_()
        else:
            flat, report = placement.probe_allocation(alloc, time_fn, nflat * 4, dev, candidates=candidates, mem_fraction=mem_fraction)
        self._flat, self._out = flat, views(flat)
        self._launchers.clear()
        self._restore(snap)
        report["buffer_MB"] = round(nflat * 4 / 1e6, 1)
        self.placement = report
        return report

    def use_device_rng_counter(self, on=True):
        """Keep the per-step offset of the device counter RNG (auto-reset draws, motor noise) in DEVICE memory
        (`FgParams.rng_offset_dev`) and advance it with a device-side add after every launch, instead of passing it
        by value.  Needed when the step loop is captured in a hipGraph (torch.cuda.CUDAGraph) with auto-reset on: by-value
        launch arguments are frozen at capture, so every replay would repeat the same reset draws.  The draws are the
        same in both modes (a captured loop, replayed, equals the loop run launch by launch)."""
        if on and self.world.rng_counter is None:
            self.world.rng_counter = torch.full((1,), int(self._rng_offset), dtype=torch.int64, device=self.world.device)
        elif not on and self.world.rng_counter is not None:
            self._rng_offset = int(self.world.rng_counter.item())
            self.world.rng_counter = None
        self._launchers.clear()
        self._roll_launchers.clear()

    def _launch_rng_offset(self):
        """By-value offset of the next launch's first step: steps taken so far + 1; with the device counter that
        count lives on the device and the by-value part is the constant 1."""
        return 1 if self.world.rng_counter is not None else self._rng_offset + 1

    def _advance_rng(self, K):
        self._rng_offset += K
        if self.world.rng_counter is not None:
            self.world.rng_counter.add_(K)             # stream-ordered after the launch that read it

    def _bound_step(self, act, rng_offset):
        """Per-step host work kept to one ctypes call: the scenario resolves every pointer and the
        FgParams struct once (`bind_step`), keyed by everything the binding depends on - action
        buffer, output buffers, stream, auto-reset flag and the world's physics constants - so a
        change of any of them simply binds again.  Scenarios without `bind_step` take the generic
        path.  The world's signature covers every agent's attributes; it is cached behind a write counter
        (core._version), so the per-step key costs a few comparisons whatever the agent count."""
        bind = getattr(self.scenario, "bind_step", None)
        if bind is None:
            return False
        key = (act.data_ptr(), self.auto_reset, _native.current_stream_fast(self.world.device),
               self.world.params_signature(), getattr(self.scenario, "_seed", 0))
        launch = self._launchers.get(key)
        if launch is None:
            if len(self._launchers) >= 64:
                self._launchers.clear()
            launch = self._launchers[key] = bind(self.world, act, self._out, auto_reset=self.auto_reset)
        launch(rng_offset)
        self.scenario._cache = self._out
        return True

    def _action_mode(self):
        """Which branch of _set_action applies (environment.py:194-216); 0 = plain continuous."""
        if self.discrete_action_input:
            return _native.FG_ACT_INDEX
        if self.discrete_action_space:
            return _native.FG_ACT_ONEHOT5
        if self.force_discrete_action:
            return _native.FG_ACT_ARGMAX
        return 0

    def _decode_action_seq(self, action_seq, mode):
        """A K-step sequence in one of the discrete action modes -> raw u [K, B, N, 2] (a staging tensor of the env, re-used
        while K stays the same).  Shapes per mode as in `step`: [K,B,N,5] floats (one-hot-5 logits), [K,B,N] indices,
        [K,B,N,2] floats (force_discrete_action: the caller's array is rewritten as the one-hot, environment.py:213-215)."""
        B, N = self.num_envs, self.num_agents
        dev = self.world.device
        tail, dtype = {_native.FG_ACT_ONEHOT5: ((B, N, 5), torch.float32),
                       _native.FG_ACT_INDEX: ((B, N), torch.int32),
                       _native.FG_ACT_ARGMAX: ((B, N, 2), torch.float32)}[mode]
        if not torch.is_tensor(action_seq) or tuple(action_seq.shape[1:]) != tail or action_seq.dim() != len(tail) + 1:
            raise ValueError("action_seq must be a tensor of shape [K%s] in this action mode" % "".join(", %d" % d for d in tail))
        K = int(action_seq.shape[0])
        src = action_seq
        if src.dtype != dtype or src.device != dev or not src.is_contiguous():
            src = src.to(device=dev, dtype=dtype).contiguous()
        stage = getattr(self, "_act_seq_stage", None)
        if stage is None or stage.shape[0] != K:
            stage = self._act_seq_stage = torch.empty((K, B, N, 2), dtype=torch.float32, device=dev)
        if K:
            _native.check(_native.load().fg_decode_actions(mode, K * B * N, src.data_ptr(), stage.data_ptr(),
                                                           _native.current_stream(dev)))
        if mode == _native.FG_ACT_ARGMAX and src is not action_seq:
            action_seq.copy_(src)                      # the reference overwrites the caller's array (:213-215)
        return stage

    def _decode_actions(self, action_n, mode, batched):
        """Non-default action modes: stage the caller's actions on the device ([B,N,5] floats,
        [B,N] indices or [B,N,2] floats) and decode them to raw u with `fg_decode_actions`."""
        B, N = self.num_envs, self.num_agents
        dev = self.world.device
        shape, dtype = {_native.FG_ACT_ONEHOT5: ((B, N, 5), torch.float32),
                        _native.FG_ACT_INDEX: ((B, N), torch.int32),
                        _native.FG_ACT_ARGMAX: ((B, N, 2), torch.float32)}[mode]
        if batched:
            if tuple(action_n.shape) != shape:
                raise ValueError("batched action must have shape %s in this action mode, got %s"
                                 % (shape, tuple(action_n.shape)))
            src = action_n
            if src.dtype != dtype or src.device != dev or not src.is_contiguous():
                src = src.to(device=dev, dtype=dtype).contiguous()
        else:
            if B != 1:
                raise ValueError("a list of per-agent actions needs num_envs == 1; pass a batched tensor")
            if len(action_n) != N:
                raise ValueError("expected %d agent actions, got %d" % (N, len(action_n)))
            host = np.asarray([np.asarray(a) for a in action_n]).reshape((1,) + shape[1:])
            src = torch.as_tensor(host).to(device=dev, dtype=dtype).contiguous()
        _native.check(_native.load().fg_decode_actions(mode, B * N, src.data_ptr(), self._act.data_ptr(),
                                                       _native.current_stream(self.world.device)))
        if mode == _native.FG_ACT_ARGMAX:
            if batched:
                if src is not action_n:
                    action_n.copy_(src)           # the reference overwrites the caller's array (:213-215)
            else:
                sens = [a.accel if a.accel is not None else 5.0 for a in self.agents]
                onehot = src[0].cpu().numpy()
                for i, a in enumerate(action_n):  # ... and scales it through the `u` view (:216,:221)
                    if isinstance(a, np.ndarray):
                        a[:] = 0.0
                        a[0:2] = onehot[i] * sens[i]
        return self._act

    def _stage_reference_actions(self, action_n):
        """environment.py:121-122,187-236 for the continuous path, B == 1."""
        if self.num_envs != 1:
            raise ValueError("a list of per-agent actions needs num_envs == 1; pass a [B,N,2] tensor")
        if len(action_n) != self.num_agents:
            raise ValueError("expected %d agent actions, got %d" % (self.num_agents, len(action_n)))
        pinned = self._act_host is not None
        host = self._act_host.numpy() if pinned else np.empty((1, self.num_agents, 2), dtype=np.float32)
        for i, (a, agent) in enumerate(zip(action_n, self.agents)):
            if isinstance(a, (list, tuple)):
                # the reference fails at `agent.action.u *= sensitivity` (:221)
                raise TypeError("can't multiply sequence by non-int of type 'float'")
            a = a if isinstance(a, np.ndarray) else np.asarray(a)
            host[0, i] = a[0:2]
            sens = agent.accel if agent.accel is not None else 5.0
            a[0:2] *= sens            # the reference scales the caller's array in place (:216,:221)
        if pinned:         # the previous step's copy has completed: every reference-style step ends with a stream sync
            self._act.copy_(self._act_host, non_blocking=True)
        else:
            self._act.copy_(torch.from_numpy(host))
        return self._act

    def _batched_result(self):
        o = self._out
        rew = o["reward"].unsqueeze(-1)
        if not self.shared_reward:
            first = getattr(self.scenario, "first_reward", None)      # a reference-style Scenario file: environment.py:128 vs :130
            rew = (o["indiv"] if first is None else first).unsqueeze(-1)
        return o["obs"], rew, o["done"].view(torch.bool), {"individual_reward": o["indiv"]}   # 0/1 bytes: a view, no kernel

    def _reference_result(self):
        o = self._out
        N = self.num_agents
        if self._host is not None:        # one pinned device-to-host copy + one stream sync for the whole step
            self._host.copy_(self._flat, non_blocking=True)
            torch.cuda.current_stream(self._flat.device).synchronize()
            h = self._host.numpy()
            n_obs = o["obs"].numel()
            obs = h[:n_obs].reshape(N, -1).astype(np.float64)
            shared = float(h[n_obs])
            indiv = h[n_obs + N:n_obs + 2 * N].astype(np.float64)
            done = bool(h[n_obs + 2 * N:].view(np.uint8)[0])
        else:
            obs = o["obs"][0].double().cpu().numpy()
            indiv = o["indiv"][0].double().cpu().numpy()
            shared = float(o["reward"][0, 0].double().cpu())
            done = bool(o["done"][0, 0].cpu())
        obs_n = [obs[i] for i in range(N)]
        if self.shared_reward:
            reward_n = [[shared]] * N                     # :136-138
        else:
            first = getattr(self.scenario, "first_reward", None)      # see _batched_result
            per_agent = indiv if first is None else first[0].double().cpu().numpy()
            reward_n = [[float(per_agent[i])] for i in range(N)]
        done_n = [done] * N
        info_n = [{'individual_reward': float(indiv[i])} for i in range(N)]
        return obs_n, reward_n, done_n, info_n

    # ----------------------------------------------------------------- reset
    def reset(self, batched=None):
        """environment.py:144-156.  Returns a list of N float64 arrays when
        num_envs == 1 (reference style) unless batched=True; else obs [B,N,D]."""
        self.current_step = 0
        self.reset_callback(self.world)
        self.agents = self.world.policy_agents
        self.scenario.observe_batch(self.world, self._out)
        if batched is None:
            batched = self.num_envs != 1
        if batched:
            return self._out["obs"]
        obs = self._out["obs"][0].double().cpu().numpy()
        return [obs[i] for i in range(self.num_agents)]

    # ---------------------------------------------------- per-agent callbacks
    def _get_info(self, agent):
        if self.info_callback is None:
            return {}
        return self.info_callback(agent, self.world)

    def _get_obs(self, agent):
        if self.observation_callback is None:
            return np.zeros(0)
        return self.observation_callback(agent, self.world)

    def _get_done(self, agent):
        if self.done_callback is None:
            return self.current_step >= self.world_length
        return self.done_callback(agent, self.world)

    def _get_reward(self, agent):
        if self.reward_callback is None:
            return 0.0
        return self.reward_callback(agent, self.world)

    def render(self, mode='human', close=False):
        raise NotImplementedError("the pyglet renderer of the reference (environment.py:243-393) "
                                  "is out of scope for the MI355X hot path")

    def close(self):
        """Drop the env's own placed buffers and bound launches: their arenas go back to the driver with the last tensor."""
        self._auto_out.clear()
        self._roll_launchers.clear()
        self._launchers.clear()
        self.scenario._cache = None
