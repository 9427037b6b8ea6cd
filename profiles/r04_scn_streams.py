#!/usr/bin/env python3
"""Do the small output streams (reward, individual reward, done: 27 bytes per env-step in three arrays) hold back the
observation stream of the lane kernels?  basic_formation_env 3 x 65536 x 80 and formation_hd_partial_range_env 4 x 65536 x 60
through the C ABI with and without them, same observation buffer."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd"), os.path.join(ROOT, "tests")]
import torch                                              # noqa: E402
from formation_gym import _native                         # noqa: E402
import test_gpu_scenario_lane as T                        # noqa: E402

lib = _native.load()
B = 65536
for shape, K in ((T.SHAPES[0], 80), (T.SHAPES[2], 60)):
    kind, N, L, M = shape[:4]
    st, p, sc, D = T._setup(*shape, B=B, crowd=1.0, seed=3)
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    acts = (torch.rand((K, B, N, 2), generator=gen, device="cuda") * 2 - 1).contiguous()
    f = dict(dtype=torch.float32, device="cuda")
    obs = torch.empty((K, B, N, D), **f)
    rew, ind = torch.empty((K, B, N), **f), torch.empty((K, B, N), **f)
    done = torch.zeros((K, B, N), dtype=torch.uint8, device="cuda")
    sc.variant = 0

    def run(small, reps=6):
        s = {k: v.clone() for k, v in st.items()}
        args = lambda: (p, sc, B, N, K, s["px"].data_ptr(), s["py"].data_ptr(), s["vx"].data_ptr(), s["vy"].data_ptr(), acts.data_ptr(),
                        s["lm"].data_ptr(), _native.ptr(s.get("opos")), _native.ptr(s.get("ovel")), s["step"].data_ptr(), obs.data_ptr(),
                        rew.data_ptr(), ind.data_ptr() if small else None, done.data_ptr() if small else None,
                        None, 1, None)
        _native.check(lib.fg_rollout_scenario(*args()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            _native.check(lib.fg_rollout_scenario(*args()))
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps / K * 1e3
    for _ in range(3):
        run(True, 2)
    res = {True: [], False: []}
    for _ in range(5):
        for small in (True, False):
            res[small].append(run(small))
    for small in (True, False):
        us = sorted(res[small])[2]
        byts = B * N * D * 4 + B * N * 12 + (B * N * 5 if small else 0)
        print("%s %d x %d x %d, %s: %.2f us/step, observation %.2f TB/s, all bytes %.2f TB/s" % (
            kind, N, B, K, "all outputs" if small else "observations + shared reward only", us, B * N * D * 4 / us / 1e6, byts / us / 1e6), flush=True)
