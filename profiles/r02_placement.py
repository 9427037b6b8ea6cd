#!/usr/bin/env python3
"""Does the rate of the K-step rollout launch depend on WHERE its observation buffer lies?
(a) the same allocation at different byte offsets, (b) fresh allocations.   python profiles/r02_placement.py N B K"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
import formation_gym                                      # noqa: E402

N, B, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = "cuda:0"
env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device=dev)
env.scenario.reset_device(env.world, rng_offset=3)
env.auto_reset = True
acts = torch.rand((K, B, N, 2), device=dev) * 2 - 1
small = dict(reward=torch.empty((K, B, N), device=dev), indiv=torch.empty((K, B, N), device=dev),
             done=torch.zeros((K, B, N), dtype=torch.uint8, device=dev))
nfl = K * B * N * 6 * N
bytes_launch = (24 * N * N + 53 * N + 16) * B * K


def time_buffer(obs, reps=12):
    out = dict(small, obs=obs)
    for _ in range(3):
        env.rollout(acts, out=out)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for r in range(reps):
        env.rollout(acts, out=out)
        e[r + 1].record()
    torch.cuda.synchronize()
    ms = sorted(e[r].elapsed_time(e[r + 1]) for r in range(reps))
    return ms[len(ms) // 2]


print("N=%d B=%d K=%d  buffer %.1f MB" % (N, B, K, nfl * 4 / 1e6))
slack = 64 << 20
big = torch.empty(nfl + slack // 4, dtype=torch.float32, device=dev)
print("(a) one allocation (base %#x), observation buffer at byte offset:" % big.data_ptr())
for off in [0, 16, 128, 4096, 65536, 1 << 18, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 6 << 20, 8 << 20, 12 << 20, 16 << 20, 24 << 20, 32 << 20, 48 << 20,
            (1 << 20) + 65536, (5 << 20) + 4096 * 3]:
    obs = big[off // 4: off // 4 + nfl].view(K, B, N, 6 * N)
    ms = time_buffer(obs)
    print("  offset %10d  %.3f ms  %.0f GB/s" % (off, ms, bytes_launch / (ms * 1e-3) / 1e9))
del big
torch.cuda.empty_cache()
print("(b) fresh allocations (each held while the next is made, so they land on different pages):")
keep = []
for i in range(8):
    buf = torch.empty(nfl, dtype=torch.float32, device=dev)
    keep.append(buf)
    ms = time_buffer(buf.view(K, B, N, 6 * N))
    print("  allocation %d at %#x  %.3f ms  %.0f GB/s" % (i, buf.data_ptr(), ms, bytes_launch / (ms * 1e-3) / 1e9))

# (c) physically contiguous allocations (hipExtMallocWithFlags + hipDeviceMallocContiguous), wrapped through
#     __cuda_array_interface__
import ctypes                                              # noqa: E402
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipFree.argtypes = [ctypes.c_void_p]


class Ext(object):
    def __init__(self, nfloats, flags):
        self.ptr = ctypes.c_void_p()
        rc = hip.hipExtMallocWithFlags(ctypes.byref(self.ptr), nfloats * 4, flags)
        assert rc == 0, "hipExtMallocWithFlags rc=%d" % rc
        self.__cuda_array_interface__ = {"shape": (nfloats,), "typestr": "<f4", "data": (self.ptr.value, False), "version": 2}


del keep
torch.cuda.empty_cache()
print("(c) hipExtMallocWithFlags, flag 0x4 = hipDeviceMallocContiguous (flag 0 for comparison):")
held = []
for i, flags in enumerate([4, 0, 4, 0, 4, 4]):
    try:
        e = Ext(nfl, flags)
    except AssertionError as ex:
        print("  allocation failed:", ex)
        continue
    held.append(e)
    t = torch.as_tensor(e, device=dev)
    ms = time_buffer(t.view(K, B, N, 6 * N))
    print("  flags %d at %#x  %.3f ms  %.0f GB/s" % (flags, e.ptr.value, ms, bytes_launch / (ms * 1e-3) / 1e9))
