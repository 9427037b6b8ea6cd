"""The batched-callback contract for user scenarios: a Scenario written on DEVICE TENSORS.

The reference's plugin API (formation_gym/scenario.py:4-12; call sites environment.py:113-184) hands a scenario one agent
of one world at a time: `observation(agent, world)`, `reward(agent, world)`, NumPy vectors in `entity.state`.  A file
written that way still loads here (callback_scenario.py) but pays a device -> host copy and B x N Python calls per step; its
episode restarts and K-step calls run on the host, step by step.  A scenario that wants the whole batch on the
GPU without writing a kernel derives from `TensorScenario` instead and states the same four things on tensors:

    class Scenario(TensorScenario):
        def build_world(self, world, num_agents, **kwargs):   # what make_world does in the reference: append Agents /
            ...                                               #   Landmarks to `world`, set its constants (core.py:113-139)
        def reset_batch(self, world, mask):                   # reset_world for the envs of `mask` (bool [B] on the device,
            ...                                               #   None = every env): write the state, draw from self.generator
        def observation_batch(self, world):                   # -> float32 [B, N, D]: every agent's observation
            ...
        def reward_batch(self, world):                        # -> float32 [B, N]: every agent's own reward
            ...

State lives where the kernels keep it: `world.pos_x / pos_y / vel_x / vel_y` [B, N], `world.landmark_pos` [B, L, 2],
`world.step_count` [B] (`world.get_state()` stacks positions / velocities to [B, N, 2]; `world.set_state(pos, vel, mask=)`
writes them back for the envs of a mask).  Per-env attributes of the scenario (a target shape, a radius) are tensors with a
leading B.  `_set_action` + `World.step` - the O(N^2) part - stay one HIP launch for all envs (`fg_physics_step`, with
everything it honours: per-agent mass / size / accel / max_speed, walls, immovable / non-colliding / ghost agents); the
callbacks above run between launches as ordinary stream-ordered torch code, so a step needs no host synchronisation.

What the base class builds from them (the protocol `MultiAgentEnv` / `FormationVecEnv` drive, scenario.py):
    step_batch      physics launch, reward_batch, done, [auto-reset: reset_batch(finished envs)], observation_batch
    rollout_batch   K of those per call into [K, ...] slabs (`env.rollout`), episodes restarting inside the call
    observe_batch, reset_world(world, env_mask), seed, obs_dim, and the reference's per-agent `observation` / `reward`
    (views [B, D] / [B] of the batched results) for code that walks agents through the World API.
Order inside a step: rewards, then done, then (auto-reset) the reset, then observations - the observation returned with a
finished episode's reward is the RESET observation, as the vec-env worker does it (env_wrappers.py:14-18).  `reward_batch`
may write state (the reference's formation_hd_obs_env re-arms its obstacles there, :84-89); it must not change what
`observation_batch` of the SAME step reads for some agents only - the reference interleaves the two per agent
(environment.py:126-134), a batch cannot.
"""
import numpy as np
import torch

from .core import World
from .scenario import BaseScenario


class TensorScenario(BaseScenario):
    PATH = ("batched tensor callbacks (TensorScenario): _set_action + World.step in one HIP launch, the scenario's "
            "observation / reward / reset on device tensors - no host round trip")
    DEFAULT_PLACED = False        # env.rollout(out=None): ordinary tensors (the placement probe times HBM-bound kernels; these
                                  # steps are launch-bound torch code)

    def __init__(self):
        self._seed = 1
        self.generator = None     # torch.Generator on the world's device, made by make_world, re-seeded by seed()
        self._streams = None

    # ---- what a scenario file defines -----------------------------------------------------------
    def build_world(self, world, num_agents, **kwargs):
        raise NotImplementedError()

    def reset_batch(self, world, mask):
        raise NotImplementedError()

    def observation_batch(self, world):
        raise NotImplementedError()

    def reward_batch(self, world):
        raise NotImplementedError()

    def done_batch(self, world):
        """environment.py:172-178 without a done_callback: every agent of an env is done from step world_length on."""
        return (world.step_count >= int(world.world_length))[:, None].expand(world.num_envs, len(world.policy_agents))

    # ---- construction / seeding -------------------------------------------------------------------
    def make_world(self, num_agents=3, num_envs=1, device=None, **kwargs):
        world = World(num_envs=num_envs, device=device)
        world.collaborative = True
        self.build_world(world, num_agents, **kwargs)
        world.allocate()
        world.scenario = self
        self.generator = torch.Generator(device=world.device)
        self.generator.manual_seed(self._seed)
        self.reset_world(world)
        return world

    def seed(self, seed=None):
        """environment.py:106-110 (default seed 1): re-seeds the device generator and the per-env host streams."""
        self._seed = 1 if seed is None else int(seed)
        if self.generator is not None:
            self.generator.manual_seed(self._seed)
        self._streams = None

    def numpy_streams(self, world):
        """One legacy MT19937 stream per env, seeded seed + 1000 * (global env index) - the convention of the reference's
        workers (train/maddpg-v2/main.py:19-30).  For scenarios that want `reset_batch` to draw exactly what the reference's
        `reset_world` would (bit-exact episodes, at the price of a host round trip per reset)."""
        if self._streams is None or len(self._streams) != world.num_envs:
            base = int(getattr(self, "env_base", 0))
            self._streams = [np.random.RandomState(self._seed + 1000 * (base + b)) for b in range(world.num_envs)]
        return self._streams

    # ---- the batched protocol ---------------------------------------------------------------------
    def reset_world(self, world, env_mask=None):
        mask = None
        if env_mask is not None:
            mask = torch.as_tensor(np.asarray(env_mask, dtype=bool), device=world.device)
        self.reset_batch(world, mask)
        if mask is None:
            world.step_count.zero_()
        else:
            world.step_count.masked_fill_(mask, 0)
        world.state_version += 1

    def obs_dim(self, world):
        return int(self.observation_batch(world).shape[-1])

    def _emit(self, world, out, rewards, auto_reset):
        B, N = world.num_envs, len(world.policy_agents)
        if rewards:
            indiv = self.reward_batch(world)
            if tuple(indiv.shape) != (B, N):
                raise ValueError("reward_batch must return [B, N] = %s, got %s" % ((B, N), tuple(indiv.shape)))
            if out.get("indiv") is not None:
                out["indiv"].copy_(indiv)
            if out.get("reward") is not None:            # environment.py:136-138: the shared reward is the sum over agents
                out["reward"].copy_(indiv.double().sum(1, keepdim=True).float().expand(B, N))
        done = None
        if out.get("done") is not None or auto_reset:
            done = self.done_batch(world)
            if out.get("done") is not None:
                out["done"].copy_(done)
        if auto_reset:                                    # env_wrappers.py:14-18: `if all(done): ob = env.reset()`
            mask = done.all(dim=1)
            self.reset_batch(world, mask)
            world.step_count.masked_fill_(mask, 0)
        if out.get("obs") is not None:
            obs = self.observation_batch(world)
            if tuple(obs.shape) != tuple(out["obs"].shape):
                raise ValueError("observation_batch must return %s, got %s" % (tuple(out["obs"].shape), tuple(obs.shape)))
            out["obs"].copy_(obs)

    def _physics(self, world, act, rng_offset):
        if act.data_ptr() != world.action_u.data_ptr():
            world.action_u.copy_(act)
        world.step(rng_offset=rng_offset)                 # _set_action's scaling + World.step: one launch (fg_physics_step)
        world.world_step -= 1                             # MultiAgentEnv counts the steps itself
        world.step_count.add_(1)

    def step_batch(self, world, act, out, auto_reset=False, rng_offset=0):
        self._physics(world, act, rng_offset)
        self._emit(world, out, True, auto_reset)

    def bind_step(self, world, act, out, auto_reset=False):
        """`launch(rng_offset)` = this step with its buffers fixed (MultiAgentEnv._bound_step, FormationVecEnv.capture: a
        step loop captured in a hipGraph works for a tensor scenario whose callbacks do not touch the host - resets from
        `self.generator`, whose state the graph is told about)."""
        def launch(rng_offset=0):
            self.step_batch(world, act, out, auto_reset=auto_reset, rng_offset=rng_offset)
            return act, out
        return launch

    def snapshot_state(self):
        """The scenario's own per-env tensors and the generator state (MultiAgentEnv._snapshot: probes and graph capture
        leave the env untouched)."""
        tensors = {k: v.clone() for k, v in vars(self).items() if torch.is_tensor(v)}
        return tensors, None if self.generator is None else self.generator.get_state()

    def restore_state(self, snap):
        tensors, gen = snap
        for k, v in tensors.items():
            getattr(self, k).copy_(v)
        if gen is not None:
            self.generator.set_state(gen)

    def observe_batch(self, world, out):
        self._emit(world, {"obs": out.get("obs"), "done": out.get("done")}, False, False)

    def rollout_batch(self, world, act_seq, out, obs_every=1, auto_reset=False, rng_offset=0):
        """K steps per call: act_seq [K, B, N, 2]; out tensors carry a leading K (obs: K // obs_every).  The same results as K
        `step_batch` calls; nothing in between touches the host."""
        K = int(act_seq.shape[0])
        for k in range(K):
            self._physics(world, act_seq[k], rng_offset + k)
            want = (k + 1) % obs_every == 0
            self._emit(world, dict(obs=out["obs"][k // obs_every] if want and out.get("obs") is not None else None,
                                   reward=out["reward"][k] if out.get("reward") is not None else None,
                                   indiv=out["indiv"][k] if out.get("indiv") is not None else None,
                                   done=out["done"][k] if out.get("done") is not None else None), True, auto_reset)

    # ---- the reference's per-agent callbacks: views of the batched results -------------------------
    def observation(self, agent, world):
        return self.observation_batch(world)[:, agent.i]

    def reward(self, agent, world):
        return self.reward_batch(world)[:, agent.i]
