"""ring_patrol_env - an ORIGINAL scenario file written against the reference's plugin API only
(formation_gym/scenario.py:4-12: make_world / reset_world / observation / reward per agent, NumPy vectors in
`entity.state`).  It is not a copy of any file of the reference.  The same file is loaded
  * by the real reference in tests/golden/make_golden.py (fixture ring_patrol_n5.npz) and
  * by this package through formation_gym.make_env(<path>, ...) - the callback adapter -
and both must agree (tests/test_gpu_callback_plugin.py).

Task: agents of different size and mass patrol a ring of radius `self.radius` around a beacon and keep apart.
"""
import numpy as np
from formation_gym.core import World, Agent, Landmark
from formation_gym.scenario import BaseScenario


class Scenario(BaseScenario):
    def make_world(self, num_agents=5, episode_length=20):
        world = World()
        world.world_length = episode_length
        world.dim_c = 2
        world.collaborative = True
        world.agents = [Agent() for _ in range(num_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'patrol %d' % i
            agent.collide = True
            agent.silent = True
            agent.size = 0.04 + 0.015 * (i % 3)          # three sizes
            agent.initial_mass = 1.0 + 0.75 * (i % 2)    # two masses
            if i % 4 == 3:
                agent.max_speed = 0.5
        world.landmarks = [Landmark()]
        world.landmarks[0].name = 'beacon'
        world.landmarks[0].collide = False
        world.landmarks[0].movable = False
        world.landmarks[0].size = 0.02
        self.reset_world(world)
        return world

    def reset_world(self, world):
        for agent in world.agents:
            agent.state.p_pos = np.random.uniform(-0.25, +0.25, world.dim_p)    # crowded: contacts from the first step
            agent.state.p_vel = np.zeros(world.dim_p)
            agent.state.c = np.zeros(world.dim_c)
        beacon = world.landmarks[0]
        beacon.state.p_pos = np.random.uniform(-0.3, +0.3, world.dim_p)
        beacon.state.p_vel = np.zeros(world.dim_p)
        self.radius = np.random.uniform(0.3, 0.6)

    def observation(self, agent, world):
        beacon = world.landmarks[0]
        to_beacon = beacon.state.p_pos - agent.state.p_pos
        ring_error = np.sqrt(np.sum(np.square(to_beacon))) - self.radius
        others = [o for o in world.agents if o is not agent]
        rel = np.array([o.state.p_pos - agent.state.p_pos for o in others])
        order = np.argsort(np.sum(np.square(rel), axis=1), kind='stable')       # nearest neighbour first
        sizes = np.array([o.size for o in others])[order]
        return np.concatenate((agent.state.p_vel, to_beacon, [ring_error, self.radius], rel[order].flatten(), sizes))

    def reward(self, agent, world):
        beacon = world.landmarks[0]
        dist = np.sqrt(np.sum(np.square(beacon.state.p_pos - agent.state.p_pos)))
        rew = -abs(dist - self.radius)
        for other in world.agents:
            if other is agent:
                continue
            gap = np.sqrt(np.sum(np.square(other.state.p_pos - agent.state.p_pos))) - (agent.size + other.size)
            rew -= 0.1 * np.exp(-gap / 0.1)              # smooth repulsion, no thresholds: well-conditioned in fp32
        return rew

    def benchmark_data(self, agent, world):
        return {'ring_error': self.observation(agent, world)[4]}
