"""The rule by which a launch finds its device (`DeviceGuard` in csrc/formation_hip.hip), on ONE GPU.
tests/test_gpu_multidevice.py needs two visible GPUs and has never run (the pool hands out one); the rule itself - the
stream's device when there is a stream, else the device of the first state pointer, else "cannot tell" - is exported as
`fg_launch_device` and exercised here on everything a launch can be handed: torch tensors on the default stream (handle 0),
side streams, memory composed with fg_arena_map, host pointers, NULL.  With one visible GPU the entry points skip the
query altogether (nothing to switch to); what is pinned is that the answers the two-GPU path would act on are right and
that a failed query leaves no error behind for the next HIP call."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gym-formation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu


def test_launch_device_rule_on_one_gpu():
    import formation_gym
    from formation_gym import _native, placement
    lib = _native.load()
    dev = torch.device("cuda:0")
    t = torch.zeros(1024, device=dev)
    side = torch.cuda.Stream(device=dev)
    default = torch.cuda.current_stream(dev).cuda_stream
    assert default == 0                                        # torch's default stream is the NULL stream: it names no device
    assert lib.fg_launch_device(None, t.data_ptr()) == 0       # ... so the data says where the launch belongs
    assert lib.fg_launch_device(side.cuda_stream, None) == 0   # a real stream names its device
    assert lib.fg_launch_device(side.cuda_stream, t.data_ptr()) == 0
    assert lib.fg_launch_device(None, None) == -1              # nothing to go by: the current device
    host = np.zeros(16, dtype=np.float32)
    assert lib.fg_launch_device(None, host.ctypes.data) == -1  # a host pointer is nobody's device memory
    # memory composed of an arena's chunks (HIP virtual memory management): the same answer as for an ordinary allocation -
    # formation_hip.hip used to hedge ("arena pointers may carry no device attribute on every runtime"); on this stack they do
    arena = placement.Arena(64 << 20, dev, 32 << 20)
    addr = arena.map([1, 0])
    flat = arena.floats(addr, 1 << 20)
    where = lib.fg_launch_device(None, flat.data_ptr())
    where_mid = lib.fg_launch_device(None, flat.data_ptr() + (40 << 20))      # inside the second chunk of the mapping
    assert where == 0 and where_mid == 0, (where, where_mid)
    # the failed queries above left no sticky error: the next launches work, on the default and on a side stream, into
    # ordinary and into arena memory, and give the same bits
    N, B = 9, 64
    a = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
    b = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
    for e in (a, b):
        e.seed(5); e.reset()
    act = torch.rand((B, N, 2), device=dev) * 2 - 1
    o1, r1, d1, _ = a.step(act)
    obs_arena = flat[:B * N * 6 * N].view(B, N, 6 * N)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        out = dict(b._out, obs=obs_arena)
        b.scenario.step_batch(b.world, act, out)
    side.synchronize()
    assert torch.equal(o1, obs_arena) and torch.equal(r1.squeeze(-1), out["reward"])
    del flat, obs_arena, out
    arena.close()
