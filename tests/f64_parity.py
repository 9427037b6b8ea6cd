"""Test-side binding of libformation_hip_f64.so, the fp64 "parity mode" build of the fused step kernel
(gym-formation_amd/csrc/formation_hip_f64.hip: the SAME kernel source as the product library with real = double).
Test infrastructure only - the product package never loads it."""
import ctypes
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "gym-formation_amd", "lib", "libformation_hip_f64.so")


class Fg64Params(ctypes.Structure):
    _fields_ = [(k, ctypes.c_double) for k in ("dt", "damping", "contact_force", "contact_margin", "sensitivity", "mass",
                                               "dist_min", "collide_thresh")] + \
               [("world_length", ctypes.c_int32), ("reserved", ctypes.c_int32)]


_lib = None


def load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(LIB_PATH)
        lib.fg64_step_hd.restype = ctypes.c_int
        lib.fg64_step_hd.argtypes = [ctypes.POINTER(Fg64Params), ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 16
        lib.fg64_rollout_hd.restype = ctypes.c_int
        lib.fg64_rollout_hd.argtypes = [ctypes.POINTER(Fg64Params), ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 13
        _lib = lib
    return _lib


def default_params(**kw):
    """The constants in force at the BASELINE configs, in double."""
    d = dict(dt=0.1, damping=0.25, contact_force=1e2, contact_margin=1e-3, sensitivity=5.0, mass=1.0,
             dist_min=0.06, collide_thresh=0.03, world_length=100)
    d.update(kw)
    return Fg64Params(**d)


class Env64(object):
    """B envs of N agents held in fp64 device tensors; `step(act)` = one fg64_step_hd launch."""

    def __init__(self, pos, vel, ideal_shape, ideal_vel, step=None, params=None, indices=True):
        f = dict(dtype=torch.float64, device="cuda")
        pos = np.asarray(pos, dtype=np.float64)
        self.B, self.N = pos.shape[:2]
        B, N = self.B, self.N
        self.px = torch.as_tensor(np.ascontiguousarray(pos[..., 0]), **f)
        self.py = torch.as_tensor(np.ascontiguousarray(pos[..., 1]), **f)
        vel = np.asarray(vel, dtype=np.float64)
        self.vx = torch.as_tensor(np.ascontiguousarray(vel[..., 0]), **f)
        self.vy = torch.as_tensor(np.ascontiguousarray(vel[..., 1]), **f)
        self.shape = torch.as_tensor(np.array(np.broadcast_to(ideal_shape, (B, N, 2))), **f)
        self.ivel = torch.as_tensor(np.array(np.broadcast_to(ideal_vel, (B, 2))), **f)
        self.step_count = torch.as_tensor(np.zeros(B, dtype=np.int32) if step is None else np.asarray(step, dtype=np.int32)).cuda()
        self.obs = torch.empty((B, N, 6 * N), **f)
        self.reward = torch.empty((B, N), **f)
        self.indiv = torch.empty((B, N), **f)
        self.done = torch.zeros((B, N), dtype=torch.uint8, device="cuda")
        self.near_lm = torch.zeros((B, N), dtype=torch.int32, device="cuda") if indices else None
        self.near_ag = torch.zeros((B, N), dtype=torch.int32, device="cuda") if indices else None
        self.hd_idx = torch.zeros((B, 4), dtype=torch.int32, device="cuda") if indices else None
        self.params = params or default_params()

    def step(self, act):
        act = torch.as_tensor(np.ascontiguousarray(np.asarray(act, dtype=np.float64)), dtype=torch.float64, device="cuda")
        p = lambda t: None if t is None else t.data_ptr()
        rc = load().fg64_step_hd(self.params, self.B, self.N, p(self.px), p(self.py), p(self.vx), p(self.vy), p(act),
                                 p(self.shape), p(self.ivel), p(self.step_count), p(self.obs), p(self.reward),
                                 p(self.indiv), p(self.done), p(self.near_lm), p(self.near_ag), p(self.hd_idx),
                                 torch.cuda.current_stream().cuda_stream)
        assert rc == 0, "fg64_step_hd returned %d" % rc
        torch.cuda.synchronize()
        return self

    def pos(self):
        return torch.stack((self.px, self.py), -1).cpu().numpy()

    def vel(self):
        return torch.stack((self.vx, self.vy), -1).cpu().numpy()


def rollout64(g, params=None):
    """The fixture's T steps in ONE launch of the fp64 build of the PIPELINED rollout kernel (fg64_rollout_hd: rollout_kernel<9 | 27,
    ...> with real = double), free-running from the fixture's initial state.  Returns dict(pos, vel [B,N,2] after the launch;
    obs [T,B,N,6N], reward, indiv [T,B,N], done [T,B,N])."""
    acts = np.ascontiguousarray(np.asarray(g["acts"], dtype=np.float64))                # [T,B,N,2], fp32-representable values
    T, B, N = acts.shape[:3]
    env = Env64(g["pos0"], g["vel0"], g["ideal_shape"], g["ideal_vel"], indices=False)
    f = dict(dtype=torch.float64, device="cuda")
    act = torch.as_tensor(acts, **f)
    obs = torch.full((T, B, N, 6 * N), float("nan"), **f)
    rew = torch.full((T, B, N), float("nan"), **f)
    indiv = torch.full((T, B, N), float("nan"), **f)
    done = torch.full((T, B, N), 7, dtype=torch.uint8, device="cuda")
    rc = load().fg64_rollout_hd(params or env.params, B, N, T, env.px.data_ptr(), env.py.data_ptr(), env.vx.data_ptr(), env.vy.data_ptr(),
                                act.data_ptr(), env.shape.data_ptr(), env.ivel.data_ptr(), env.step_count.data_ptr(),
                                obs.data_ptr(), rew.data_ptr(), indiv.data_ptr(), done.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, "fg64_rollout_hd returned %d" % rc
    torch.cuda.synchronize()
    return dict(pos=env.pos(), vel=env.vel(), obs=obs.cpu().numpy(), reward=rew.cpu().numpy(), indiv=indiv.cpu().numpy(),
                done=done.cpu().numpy(), step=env.step_count.cpu().numpy())
