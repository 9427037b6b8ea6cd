"""drifting_rocks_env - an ORIGINAL scenario file written against the reference's plugin API only
(formation_gym/scenario.py:4-12).  It is not a copy of any file of the reference.  It exercises what ring_patrol_env does
not: LANDMARKS THAT COLLIDE - two movable rocks of their own mass and size that drift through the agents and push them
(core.py:240-262 over all entities, force ratio m_b / m_a :314-317), an immovable pillar that only pushes back (:319-321),
a beacon nobody collides with - and a reward callback that WRITES state: like the reference's formation_hd_obs_env
(:82-89) it re-arms the rocks' velocity every step.  Loaded by the real reference in tests/golden/make_golden.py
(fixture drifting_rocks_n4.npz) and by this package through formation_gym.make_env(<path>, ...).

Task: agents gather around the beacon while the rocks sweep through.
"""
import numpy as np
from formation_gym.core import World, Agent, Landmark
from formation_gym.scenario import BaseScenario


class Scenario(BaseScenario):
    def make_world(self, num_agents=4, episode_length=15):
        world = World()
        world.world_length = episode_length
        world.dim_c = 2
        world.collaborative = True
        world.agents = [Agent() for _ in range(num_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'gatherer %d' % i
            agent.collide = True
            agent.silent = True
            agent.size = 0.05 + 0.01 * (i % 2)
            agent.initial_mass = 1.0 + 0.5 * (i % 3)
        names = ['beacon', 'rock 0', 'pillar', 'rock 1']          # colliders and non-colliders interleaved
        world.landmarks = [Landmark() for _ in names]
        for l, name in zip(world.landmarks, names):
            l.name = name
            l.collide = name != 'beacon'
            l.movable = name.startswith('rock')
            l.size = {'b': 0.02, 'r': 0.12, 'p': 0.15}[name[0]]
            l.initial_mass = 3.0 if name == 'rock 0' else 1.5
        self.drift = np.array([[0.6, -0.1], [-0.5, 0.2]])
        self.reset_world(world)
        return world

    def rocks(self, world):
        return [l for l in world.landmarks if l.movable]

    def reset_world(self, world):
        for agent in world.agents:
            agent.state.p_pos = np.random.uniform(-0.3, +0.3, world.dim_p)
            agent.state.p_vel = np.zeros(world.dim_p)
            agent.state.c = np.zeros(world.dim_c)
        for l in world.landmarks:
            l.state.p_vel = np.zeros(world.dim_p)
            if l.name == 'beacon':
                l.state.p_pos = np.random.uniform(-0.2, +0.2, world.dim_p)
            elif l.name == 'pillar':
                l.state.p_pos = np.random.uniform(-0.1, +0.1, world.dim_p)
        for k, rock in enumerate(self.rocks(world)):
            side = -1.0 if k == 0 else 1.0
            rock.state.p_pos = np.array([side * np.random.uniform(0.35, 0.5), np.random.uniform(-0.2, 0.2)])
            rock.state.p_vel = self.drift[k].copy()

    def observation(self, agent, world):
        rel = [l.state.p_pos - agent.state.p_pos for l in world.landmarks]
        others = [o.state.p_pos - agent.state.p_pos for o in world.agents if o is not agent]
        rock_vel = [r.state.p_vel for r in self.rocks(world)]
        return np.concatenate([agent.state.p_vel] + rel + others + rock_vel)

    def reward(self, agent, world):
        beacon = world.landmarks[0]
        rew = -np.sqrt(np.sum(np.square(beacon.state.p_pos - agent.state.p_pos)))
        for k, rock in enumerate(self.rocks(world)):               # the callback writes state (cf. formation_hd_obs_env.py:82-89)
            inside = abs(rock.state.p_pos[0]) < 1.0
            rock.state.p_vel = self.drift[k].copy() if inside else np.zeros(world.dim_p)
        for l in world.landmarks:
            if l.collide:
                gap = np.sqrt(np.sum(np.square(l.state.p_pos - agent.state.p_pos))) - (agent.size + l.size)
                rew -= 0.2 * np.exp(-gap / 0.1)                     # smooth: well-conditioned in fp32
        return rew
