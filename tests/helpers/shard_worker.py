"""One rank of a sharded rollout (launched by torch.distributed.run from tests/test_gpu_parity.py): this rank's
contiguous slice of a GLOBAL env batch steps on the HIP path, results are gathered on rank 0 over gloo (host-side
gather, no data-path collective) and written to an .npz for the parent test to compare with a one-process run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]


def main():
    out_path, N, G, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dist.init_process_group("gloo")
    from formation_gym import sharding
    env, lo, hi = sharding.make_env_shard("formation_hd_env", N, G, seed=5)          # RANK / WORLD_SIZE / LOCAL_RANK from the launcher
    assert torch.cuda.current_device() == env.world.device.index
    env.auto_reset = True
    env.reset()
    env.world.step_count.copy_((torch.arange(lo, hi, dtype=torch.int32) * 7 % 100).to(env.world.device))
    gen = torch.Generator(); gen.manual_seed(123)
    acts = (torch.rand((K, G, N, 2), generator=gen) * 2 - 1)[:, lo:hi].contiguous().to(env.world.device)   # the global action tensor, sliced
    obs, rew, done, info = env.rollout(acts)
    res = {}
    for name, t in (("obs_last", obs[-1]), ("rew", rew[..., 0].permute(1, 0, 2).contiguous()),
                    ("done", done.permute(1, 0, 2).contiguous().to(torch.uint8)), ("pos_x", env.world.pos_x),
                    ("shape", env.scenario.ideal_shape)):
        full = sharding.gather_host(t, dst=0)
        if dist.get_rank() == 0:
            res[name] = full.numpy()
    if dist.get_rank() == 0:
        res["slices"] = np.array([sharding.env_slice(G, r, dist.get_world_size()) for r in range(dist.get_world_size())])
        np.savez(out_path, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
