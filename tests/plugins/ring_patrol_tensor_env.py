"""ring_patrol_tensor_env - the scenario of ring_patrol_env.py written against the batched-callback contract
(formation_gym/tensor_scenario.py): the same world, the same observation and reward, stated once for all envs on device
tensors.  tests/test_gpu_tensor_scenario.py holds it against the fixture the REAL reference produced from the per-agent file
(ring_patrol_n5.npz) and against the per-agent file itself running through the callback adapter.

`exact_reset = True` (default): `reset_batch` draws from each env's legacy MT19937 stream in the per-agent file's order
(bit-exact episodes; a host round trip per reset).  False: from the device generator (no host involvement at all).
"""
import numpy as np
import torch

from formation_gym.core import Agent, Landmark
from formation_gym.tensor_scenario import TensorScenario


class Scenario(TensorScenario):
    exact_reset = True

    def build_world(self, world, num_agents=5, episode_length=20):
        world.world_length = episode_length
        world.dim_c = 2
        world.collaborative = True
        world.agents = [Agent() for _ in range(num_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'patrol %d' % i
            agent.collide = True
            agent.silent = True
            agent.size = 0.04 + 0.015 * (i % 3)
            agent.initial_mass = 1.0 + 0.75 * (i % 2)
            if i % 4 == 3:
                agent.max_speed = 0.5
        world.landmarks = [Landmark()]
        world.landmarks[0].name = 'beacon'
        world.landmarks[0].collide = False
        world.landmarks[0].movable = False
        world.landmarks[0].size = 0.02
        self.radius = None                                    # [B]: one ring per env
        self._sz = self._others = None

    def reset_batch(self, world, mask):
        B, N, dev = world.num_envs, len(world.agents), world.device
        if self.radius is None:
            self.radius = torch.zeros(B, device=dev)
        if self.exact_reset:
            which = range(B) if mask is None else np.nonzero(mask.cpu().numpy())[0]
            if len(which) == 0:
                return
            pos, _ = world.get_state()
            pos = pos.cpu().numpy().astype(np.float64)
            beacon = world.landmark_pos.cpu().numpy().astype(np.float64)
            radius = self.radius.cpu().numpy().astype(np.float64)
            for b in which:                                   # the draw order of ring_patrol_env.reset_world
                rs = self.numpy_streams(world)[b]
                for i in range(N):
                    pos[b, i] = rs.uniform(-0.25, +0.25, 2)
                beacon[b, 0] = rs.uniform(-0.3, +0.3, 2)
                radius[b] = rs.uniform(0.3, 0.6)
            sel = torch.ones(B, dtype=torch.bool, device=dev) if mask is None else mask
            world.set_state(pos, torch.zeros((B, N, 2), device=dev), mask=sel)
            world.landmark_pos.copy_(torch.where(sel[:, None, None], torch.as_tensor(beacon, dtype=torch.float32).to(dev),
                                                 world.landmark_pos))
            self.radius.copy_(torch.where(sel, torch.as_tensor(radius, dtype=torch.float32).to(dev), self.radius))
            return
        u = lambda *shape: torch.rand(shape, generator=self.generator, device=dev)
        sel = torch.ones(B, dtype=torch.bool, device=dev) if mask is None else mask
        world.set_state(-0.25 + 0.5 * u(B, N, 2), torch.zeros((B, N, 2), device=dev), mask=sel)
        world.landmark_pos.copy_(torch.where(sel[:, None, None], -0.3 + 0.6 * u(B, 1, 2), world.landmark_pos))
        self.radius.copy_(torch.where(sel, 0.3 + 0.3 * u(B), self.radius))

    def _tables(self, world):
        """Agent sizes [N] and, per agent, the indices of the others in index order [N, N - 1] - made once, on the device
        (nothing below copies from the host or asks it for a size: the step can be captured in a hipGraph)."""
        if self._sz is None:
            N = len(world.agents)
            self._sz = torch.tensor([a.size for a in world.agents], dtype=torch.float32, device=world.device)
            self._others = torch.tensor([[j for j in range(N) if j != i] for i in range(N)], device=world.device)
        return self._sz, self._others

    def observation_batch(self, world):
        pos, vel = world.get_state()                          # [B, N, 2]
        B, N = pos.shape[:2]
        to_beacon = world.landmark_pos[:, :1] - pos           # [B, N, 2]
        ring_error = to_beacon.square().sum(-1).sqrt() - self.radius[:, None]
        sz, others = self._tables(world)
        rel = pos[:, others] - pos[:, :, None, :]                                               # [B, N, N - 1, 2], index order
        sizes = sz[others][None].expand(B, N, N - 1)
        order = torch.argsort(rel.square().sum(-1), dim=-1, stable=True)                       # nearest neighbour first
        rel = torch.gather(rel, 2, order[..., None].expand(B, N, N - 1, 2))
        sizes = torch.gather(sizes, 2, order)
        return torch.cat((vel, to_beacon, ring_error[..., None], self.radius[:, None, None].expand(B, N, 1),
                          rel.reshape(B, N, 2 * (N - 1)), sizes), -1)

    def reward_batch(self, world):
        pos, _ = world.get_state()
        N = pos.shape[1]
        dist = (world.landmark_pos[:, :1] - pos).square().sum(-1).sqrt()
        sz, _ = self._tables(world)
        gap = (pos[:, None, :, :] - pos[:, :, None, :]).square().sum(-1).sqrt() - (sz[:, None] + sz[None, :])[None]
        push = 0.1 * torch.exp(-gap / 0.1)
        push = push.masked_fill(torch.eye(N, dtype=torch.bool, device=pos.device)[None], 0.0)
        return -(dist - self.radius[:, None]).abs() - push.sum(-1)

    def benchmark_data(self, agent, world):
        return {'ring_error': self.observation_batch(world)[:, agent.i, 4]}
