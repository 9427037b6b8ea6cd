"""Scenario plugin base class (reference formation_gym/scenario.py:4-12).

A scenario file `formation_gym/envs/<name>.py` defines `class Scenario(BaseScenario)`.
The reference calls `observation(agent, world)` / `reward(agent, world)` once per
agent from Python; here the per-agent callbacks are views into results that ONE
fused HIP launch produced for all agents of all environments, and the batched
hooks below are what MultiAgentEnv drives.
"""


class BaseScenario(object):
    # create elements of the world
    def make_world(self):
        raise NotImplementedError()

    # create initial conditions of the world
    def reset_world(self, world):
        raise NotImplementedError()

    def info(self, agent, world):
        return {}

    # ---- batched protocol (MI355X-native) ---------------------------------
    def obs_dim(self, world):
        """Length of one agent's observation vector."""
        raise NotImplementedError()

    def step_batch(self, world, act, out):
        """_set_action + World.step + observation/reward/done for all envs (one launch)."""
        raise NotImplementedError()

    def observe_batch(self, world, out):
        """observation (+ reward/done) on the current state (one launch)."""
        raise NotImplementedError()
