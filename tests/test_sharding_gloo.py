"""World-size-2 `gloo` test (CPU) of the multi-GPU path: the env batch is cut
into contiguous slices with global per-env seeds, ranks work independently, and
only a host-side gather + a MAX of timings cross ranks.  The per-rank compute in
this test is the CPU oracle (a stand-in for the HIP kernels, which need a GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from formation_gym import sharding
from oracle import formation_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, N, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.env_slice(B, rank, world)
    st = O.reset_hd(sharding.global_seeds(1, lo, hi), N)
    acts = np.random.RandomState(0).uniform(-1, 1, (B, N, 2))[lo:hi]
    st, out = O.step_hd(st, acts)
    rew = sharding.gather_host(torch.as_tensor(out["reward"][..., 0]))
    slowest = sharding.max_over_ranks(1.0 + rank)
    dist.barrier()
    if rank == 0:
        q.put((rew.numpy(), slowest))
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [6, 7])
def test_two_rank_sharding_matches_single_process(B):
    N, world = 9, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    rew, slowest = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    st = O.reset_hd(sharding.global_seeds(1, 0, B), N)
    _, out = O.step_hd(st, np.random.RandomState(0).uniform(-1, 1, (B, N, 2)))
    np.testing.assert_array_equal(rew, out["reward"][..., 0])      # independent of the GPU count
    assert slowest == 2.0


def test_env_slices_partition_the_batch():
    for B in (1, 7, 4096, 65536):
        for world in (1, 2, 3, 8):
            spans = [sharding.env_slice(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.env_slice(8, 2, 2)
    np.testing.assert_array_equal(sharding.global_seeds(5, 2, 5), [2005, 3005, 4005])
