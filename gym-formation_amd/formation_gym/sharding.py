"""Multi-GPU sharding of the environment batch.

Environments never interact (the reference itself runs them in separate
processes: train/maddpg-v2/utils/env_wrappers.py:40-56), so the batch is cut
into contiguous slices, one process per GPU, and every rank runs the same
kernels on its slice.  No data-path collective exists; `torch.distributed` is
used only for the timing barrier / max and for an optional host-side gather of
small per-env results.  Per-env seeds follow the global env index
(`seed + 1000 * b`, train/maddpg-v2/main.py:19-30), so results do not depend on
the number of GPUs.
"""
import numpy as np
import torch
import torch.distributed as dist


def env_slice(global_envs, rank, world_size):
    """Contiguous [lo, hi) slice of the global env index range owned by `rank`
    (sizes differ by at most one)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, rem = divmod(int(global_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_seeds(seed, lo, hi):
    """Reset-stream seed of every env in [lo, hi): seed + 1000 * global_index."""
    return int(seed) + 1000 * np.arange(lo, hi, dtype=np.int64)


def make_env_shard(scenario_name, num_agents, global_envs, seed=1, rank=None, world_size=None, local_rank=None,
                   **scenario_kwargs):
    """This rank's slice of a batch of `global_envs` environments, on this rank's GPU: one process per GPU
    (rank / world size / local rank from the launcher's RANK, WORLD_SIZE, LOCAL_RANK unless given).  Env g of
    the GLOBAL batch draws its reset stream from `seed + 1000 g` wherever it lives, so results do not depend
    on the number of GPUs.  Returns (env, lo, hi): the env owns global envs [lo, hi)."""
    import os
    import formation_gym
    rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
    world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else int(world_size)
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank))) if local_rank is None else int(local_rank)
    lo, hi = env_slice(global_envs, rank, world_size)
    ndev = max(1, torch.cuda.device_count())
    device = torch.device("cuda", local_rank % ndev)
    # one process per GPU: the rank's GPU becomes the process's current device, so that everything the caller
    # allocates next to the env (action pools, rollout buffers, `torch.cuda.current_stream()`) lands on it too
    torch.cuda.set_device(device)
    env = formation_gym.make_env(scenario_name, False, num_agents, num_envs=hi - lo, device=device, **scenario_kwargs)
    env.scenario.env_base = lo                   # env b of this rank IS global env lo + b: host reset streams
    env.seed(int(seed))                          # RandomState(seed + 1000 (lo + b)), device counter RNG keyed alike
    return env, lo, hi


def max_over_ranks(value, device=None, group=None):
    """MAX of a python float over all ranks (the bench's slowest-rank time).  `group` / `device`:
    the process group to reduce over and the device its backend wants the tensor on (RCCL: the GPU)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t[0])


def gather_host(local, dst=0):
    """Host-side gather of a per-env tensor [b_local, ...] to rank `dst`, in global
    env order.  Returns the concatenation on `dst`, None elsewhere."""
    local = local.detach().cpu()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    bucket = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local, bucket, dst=dst)
    if dist.get_rank() != dst:
        return None
    return torch.cat(bucket, dim=0)
