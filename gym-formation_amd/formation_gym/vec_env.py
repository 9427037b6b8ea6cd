"""Vectorised-env adapter with the semantics RL callers of the reference get
from `SubprocVecEnv` / `DummyVecEnv` (train/maddpg-v2/utils/env_wrappers.py:9-128):
`step` auto-resets an environment whose agents are all done and returns the
RESET observation together with the pre-reset reward/done; results are stacked
`(B, N, ...)`.  There the batch is one OS process per env talking over a Pipe;
here it is the batch dimension of one fused HIP launch, and the reset happens
inside that launch (counter RNG) or, in parity mode, on the host from each env's
own legacy MT19937 stream (seed + 1000 * rank, train/maddpg-v2/main.py:19-30).
"""
import torch


class FormationVecEnv(object):
    def __init__(self, env, reset_mode="device", numpy=False, infos="dict"):
        """reset_mode:
          'device'    counter RNG inside the fused step launch (fastest; distributional parity); every scenario in envs/
          'device_mt' (every built-in scenario) the reference's own MT19937 streams continued on the GPU (bit-exact resets,
                      no host round trip: the host mirrors the step counters, which are deterministic)
          'host'      the reference's streams on the host (bit-exact; needs a device->host sync)
        numpy=True: `reset` / `step` return what the reference's vec envs return (env_wrappers.py:68-72, :113-122): NumPy float64
          obs [B,N,D] and rews [B,N,1], bool dones [B,N] (one device-to-host copy per step); default: device tensors.
        infos: 'dict' = one dict of [B,N] tensors; 'tuple' = the reference's shape, a tuple of B lists of N dicts
          {'individual_reward': float} (SubprocVecEnv.step_wait; B x N Python objects per step: for small batches)."""
        if infos not in ("dict", "tuple"):
            raise ValueError("infos must be 'dict' or 'tuple'")
        if reset_mode not in ("device", "device_mt", "host"):
            raise ValueError("reset_mode must be 'device', 'device_mt' or 'host'")
        if reset_mode == "device_mt" and not hasattr(env.scenario, "bind_reset_mt_done"):
            raise NotImplementedError("reset_mode='device_mt' (the legacy MT19937 streams continued on the GPU) is built for "
                                      "formation_hd_env; %s resets with 'device' (counter RNG) or 'host' (bit-exact)"
                                      % type(env.scenario).__module__.rsplit(".", 1)[-1])
        self.env = env
        self.reset_mode = reset_mode
        self.num_envs = env.num_envs
        self.num_agents = env.num_agents
        self.observation_space = env.observation_space
        self.action_space = env.action_space
        if all(hasattr(a, 'adversary') for a in env.agents):    # env_wrappers.py:29-34, :102-106
            self.agent_types = ['adversary' if a.adversary else 'agent' for a in env.agents]
        else:
            self.agent_types = ['agent' for _ in env.agents]
        env.auto_reset = reset_mode == "device"
        self.numpy, self.infos = bool(numpy), infos
        import numpy as np
        # steps since each env's last reset (DummyVecEnv.ts, env_wrappers.py:108, :116-120): a host mirror of the device's
        # step counters - they are deterministic (+1 per step, 0 at world_length), so no read-back is needed after the first
        self.ts = np.zeros(self.num_envs, dtype='int')
        self._ts_synced = False
        self._mt_launch = None

    def _to_numpy(self, obs, rew, done, info):
        if self.infos == "tuple":
            indiv = info["individual_reward"].double().cpu().numpy()
            info = tuple([{'individual_reward': float(r)} for r in row] for row in indiv)
        if not self.numpy:
            return obs, rew, done, info
        return obs.double().cpu().numpy(), rew.double().cpu().numpy(), done.cpu().numpy(), info

    def get_spaces(self):
        return self.observation_space, self.action_space

    def reset(self):
        self.ts[:] = 0
        self._ts_synced = True
        if self.reset_mode == "device_mt":
            sc, world = self.env.scenario, self.env.world
            sc.upload_mt_streams(world)
            sc.reset_mt(world)                                    # all envs, on device
            self.env.current_step = 0
            sc.observe_batch(world, {"obs": self.env._out["obs"]})
            obs = self.env._out["obs"]
        else:
            obs = self.env.reset(batched=True)
        return obs.double().cpu().numpy() if self.numpy else obs

    def reset_task(self):
        """SubprocVecEnv.reset_task (env_wrappers.py:79-82) asks every worker for `env.reset_task()`, which the reference's
        MultiAgentEnv does not have: the call fails there, and it fails the same way here."""
        raise AttributeError("'MultiAgentEnv' object has no attribute 'reset_task'")

    def step(self, actions):
        """actions [B, N, 2] (a tensor, or anything torch.as_tensor takes) -> obs [B,N,D], rews [B,N,1], dones [B,N] (bool),
        infos; `ts` counts the steps since each env's last reset."""
        if not torch.is_tensor(actions):
            import numpy as np
            actions = torch.as_tensor(np.asarray(actions, dtype=np.float32), device=self.env._act.device)
        if not self._ts_synced:                                   # the env was stepped / loaded behind our back: read the counters once
            self.ts[:] = self.env.world.step_count.cpu().numpy()
            self._ts_synced = True
        obs, rew, done, info = self.env.step(actions)
        self.ts += 1
        finished = self.ts >= int(self.env.world.world_length)
        self.ts[finished] = 0
        if self.reset_mode == "device_mt":
            if finished.any():
                # ONE launch, decided on the device (step counter >= world_length): the finished envs restart from their
                # own MT19937 streams and their reset observation replaces the step's; no mask upload, no second pass over
                # the batch.  reward / done / info keep their pre-reset values.  The host mirror of the (deterministic)
                # step counters only serves to skip the launch on steps in which nobody finishes.
                if self._mt_launch is None or self._mt_launch[0] != self.env._out["obs"].data_ptr():
                    self._mt_launch = (self.env._out["obs"].data_ptr(),
                                       self.env.scenario.bind_reset_mt_done(self.env.world, self.env._out["obs"]))
                self._mt_launch[1]()
            return self._to_numpy(self.env._out["obs"], rew, done, info)
        if self.reset_mode == "host":
            mask = done.all(dim=1)
            if bool(mask.any()):
                # keep the pre-reset reward/done; replace obs of finished envs by reset obs
                rew, done = rew.clone(), done.clone()
                info = {k: v.clone() for k, v in info.items()}
                self.env.scenario.reset_world(self.env.world, env_mask=mask.cpu().numpy())
                self.env.scenario.observe_batch(self.env.world, {"obs": self.env._out["obs"]})
                obs = self.env._out["obs"]
        return self._to_numpy(obs, rew, done, info)

    def rollout(self, action_seq, out=None, obs_every=1):
        """K vec-env steps in ONE launch (`env.rollout`): action_seq [K,B,N,2] -> obs [K//obs_every,B,N,D],
        rews [K,B,N,1], dones [K,B,N], infos.  Episodes that end inside the launch restart on the device
        exactly as K `step` calls would ('device' reset mode only: the other modes reset between launches)."""
        if self.reset_mode != "device":
            raise NotImplementedError("multi-step launches reset on the device: use reset_mode='device' or call step()")
        self._ts_synced = False                                   # (re-read after a multi-step launch)
        return self.env.rollout(action_seq, out=out, obs_every=obs_every)

    def rollout_policy(self, K, num_agents_per_layer=3, out=None, obs_every=1):
        """K vec-env steps driven by the reference's built-in controller (`env.rollout_policy`): the demo loop of
        test.py:17-27 in one call; infos carries the actions taken ('device' reset mode only)."""
        if self.reset_mode != "device":
            raise NotImplementedError("multi-step launches reset on the device: use reset_mode='device' or call step()")
        self._ts_synced = False
        return self.env.rollout_policy(K, num_agents_per_layer, out=out, obs_every=obs_every)

    def capture(self, policy_fn, steps_per_replay):
        """The caller's step loop (train/maddpg-v2/main.py:77-91: policy forward -> env.step, K times) captured ONCE
        in a hipGraph and replayed: `loop = venv.capture(policy_fn, T)`, then every `loop.replay()` advances all envs by
        T steps - `act = policy_fn(obs); obs, rew, done, info = venv.step(act)` T times - at the cost of one graph launch
        instead of T x (policy kernels + step launch) host round trips (2-4x on launch-bound batches).
          policy_fn(obs [B,N,D]) -> actions [B,N,2]: device-side work only (torch modules, formation_gym.get_action_BFS,
              ...), no host synchronisation, the same shapes every call - what hipGraph capture asks of any code.  If it
              has an `out` parameter it is called as policy_fn(obs, out=slot) and may write the actions there directly.
        Episodes restart inside the graph ('device' reset mode; the counter RNG's per-step offset lives in device memory,
        so every replay draws new reset states - those of the same steps taken launch by launch).  Capturing leaves the
        env's state untouched (the warm-up pass is rolled back).  Returns a `CapturedLoop`."""
        return CapturedLoop(self, policy_fn, int(steps_per_replay))

    def step_async(self, actions):
        self._pending = self.step(actions)

    def step_wait(self):
        out, self._pending = self._pending, None
        return out

    def close(self):
        self.env.close()


class CapturedLoop(object):
    """A T-step policy-in-the-loop rollout as one replayable hipGraph (see `FormationVecEnv.capture`).

    Every step writes straight into its slot of fixed [T, ...] buffers (no copies inside the graph):
        obs [T,B,N,D], reward [T,B,N,1], done [T,B,N] (bool), info {'individual_reward' [T,B,N], 'actions' [T,B,N,2]}
    which `replay()` returns as views; the next replay overwrites them.  The observation step 0 of a replay acts on is
    the last one of the previous replay (slot T-1), seeded at capture time with the env's current observation."""

    def __init__(self, venv, policy_fn, steps):
        if venv.reset_mode != "device":
            raise NotImplementedError("a captured loop resets on the device: use reset_mode='device'")
        if steps < 1:
            raise ValueError("steps_per_replay must be >= 1")
        env = venv.env
        if getattr(env.scenario, "bind_step", None) is None or env._action_mode() or env.post_step_callback is not None:
            raise NotImplementedError("capture needs a batched scenario with continuous actions and no host callbacks")
        self.venv, self.env, self.steps, self.policy_fn = venv, env, steps, policy_fn
        try:                                       # policy_fn(obs, out=slot): an optional `out` saves a copy per step
            import inspect
            self._takes_out = "out" in inspect.signature(policy_fn).parameters
        except (TypeError, ValueError):
            self._takes_out = False
        B, N = env.num_envs, env.num_agents
        D = env._out["obs"].shape[-1]
        dev = env._act.device
        f = dict(dtype=torch.float32, device=dev)
        T = steps
        env.use_device_rng_counter(True)           # by-value launch arguments are frozen in a graph
        self.buf = dict(obs=torch.empty((T, B, N, D), **f), reward=torch.empty((T, B, N), **f),
                        indiv=torch.empty((T, B, N), **f), done=torch.zeros((T, B, N), dtype=torch.uint8, device=dev),
                        act=torch.empty((T, B, N, 2), **f))
        self.buf["obs"][T - 1].copy_(env._out["obs"])                   # what step 0 of the first replay acts on
        snap = env._snapshot()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            # launches are bound to the capture stream, one per slot (pointers resolved once)
            self._launch = [env.scenario.bind_step(env.world, self.buf["act"][t],
                                                   {k: self.buf[k][t] for k in ("obs", "reward", "indiv", "done")},
                                                   auto_reset=env.auto_reset) for t in range(T)]
            self._body()                                                # warm-up: LDS opt-ins, lazy inits, allocator pools
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        env._restore(snap)
        self.buf["obs"][T - 1].copy_(env._out["obs"])
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        gen = getattr(env.scenario, "generator", None)                  # a tensor scenario's resets draw from its own generator:
        if gen is not None:                                             # the graph advances its offset with every replay
            self.graph.register_generator_state(gen)
        with torch.cuda.graph(self.graph, stream=side):
            self._body()
        env._restore(snap)                                              # capture executes nothing; host counters did move
        self.buf["obs"][T - 1].copy_(env._out["obs"])
        self.replays = 0

    def _body(self):
        env, T = self.env, self.steps
        obs = self.buf["obs"][T - 1]
        for t in range(T):
            slot = self.buf["act"][t]
            if self._takes_out:                                          # the policy writes its actions straight into the slot
                act = self.policy_fn(obs, out=slot)
                if act is not None and act.data_ptr() != slot.data_ptr():
                    slot.copy_(act)
            else:
                slot.copy_(self.policy_fn(obs))
            self._launch[t](1 + t)                                       # by-value offset of step t + the device counter,
            obs = self.buf["obs"][t]
        env.world.rng_counter.add_(T)                                    # which advances once per replay

    def replay(self):
        env, T = self.env, self.steps
        self.graph.replay()
        self.venv._ts_synced = False                                     # `ts` is re-read from the device when next asked for
        env._rng_offset += T
        env.current_step += T
        env.world.world_step += T
        env.scenario._cache = None
        self.replays += 1
        b = self.buf
        rew = b["reward"] if env.shared_reward else b["indiv"]
        return b["obs"], rew.unsqueeze(-1), b["done"].view(torch.bool), {"individual_reward": b["indiv"], "actions": b["act"]}
