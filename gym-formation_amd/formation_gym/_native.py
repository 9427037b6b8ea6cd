"""ctypes binding of libformation_hip.so (C ABI: include/formation_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a GPU
entry point fails, an exception is raised.  Pointers handed to the library are
`tensor.data_ptr()` of PyTorch-ROCm tensors; launches go to torch's current HIP
stream so torch events and stream semantics apply.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_PKG_ROOT, "lib", "libformation_hip.so")
BUILD_SCRIPT = os.path.join(_PKG_ROOT, "csrc", "build.sh")

ABI_VERSION = 8
MAX_WALLS = 4
AGENT_PROPS = 8                 # floats per row of FgParams.agent_props: mass, size, accel, max_speed, u_noise, c_noise, flags, 0
AGENT_IMMOVABLE, AGENT_NO_COLLIDE, AGENT_GHOST, AGENT_SCRIPTED = 1, 2, 4, 8         # the flags column

FG_OK = 0
FG_ERR_BAD_ARG = -1
FG_ERR_UNSUPPORTED_N = -2
FG_ERR_ALIGNMENT = -3
FG_ERR_HIP = -4


class FgWall(ctypes.Structure):
    """Mirror of `struct FgWall`."""
    _fields_ = [("vertical", ctypes.c_int32), ("axis_pos", ctypes.c_float), ("end0", ctypes.c_float),
                ("end1", ctypes.c_float), ("width", ctypes.c_float), ("soft", ctypes.c_int32)]


class FgParams(ctypes.Structure):
    """Mirror of `struct FgParams` (include/formation_hip.h)."""
    _fields_ = [
        ("dt", ctypes.c_float),
        ("damping", ctypes.c_float),
        ("contact_force", ctypes.c_float),
        ("contact_margin", ctypes.c_float),
        ("sensitivity", ctypes.c_float),
        ("mass", ctypes.c_float),
        ("dist_min", ctypes.c_float),
        ("collide_thresh", ctypes.c_float),
        ("world_length", ctypes.c_int32),
        ("auto_reset", ctypes.c_int32),
        ("seed", ctypes.c_uint64),
        ("rng_offset", ctypes.c_uint64),
        ("accel", ctypes.c_float),
        ("max_speed", ctypes.c_float),
        ("u_noise", ctypes.c_float),
        ("num_walls", ctypes.c_int32),
        ("walls", FgWall * 4),
        ("obs_env_pitch", ctypes.c_int32),
        ("env_index_base", ctypes.c_int32),
        ("rng_offset_dev", ctypes.c_void_p),
        ("agent_props", ctypes.c_void_p),       # device float [N][AGENT_PROPS] or NULL (uniform agents)
        ("comm_state", ctypes.c_void_p),        # device float [B][N][2] or NULL (silent agents)
        ("obs_placed", ctypes.c_int32),         # 1: the observation buffer was composed with fg_arena_* (placement.py)
        ("reserved0", ctypes.c_int32),
    ]


FG_SCN_BASIC, FG_SCN_PARTIAL, FG_SCN_RANGE, FG_SCN_OBSTACLE = 1, 2, 3, 4
FG_ACT_ONEHOT5, FG_ACT_INDEX, FG_ACT_ARGMAX = 1, 2, 3          # fg_decode_actions modes


class FgScenario(ctypes.Structure):
    """Mirror of `struct FgScenario` (include/formation_hip.h)."""
    _fields_ = [
        ("kind", ctypes.c_int32),
        ("num_landmarks", ctypes.c_int32),
        ("num_obstacles", ctypes.c_int32),
        ("num_obs", ctypes.c_int32),
        ("obs_range", ctypes.c_float),
        ("obstacle_size", ctypes.c_float),
        ("obstacle_vx", ctypes.c_float),
        ("obstacle_vy", ctypes.c_float),
        ("obstacle_floor", ctypes.c_float),
        ("penalty", ctypes.c_float),
        ("variant", ctypes.c_int32),            # 0 = library's choice, 1 = the run-time-count kernel
        ("reserved", ctypes.c_int32),
    ]


class FormationHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libformation_hip: status %d: %s" % (code, msg))
        self.code = code


_P = ctypes.c_void_p
_I = ctypes.c_int
_PP = ctypes.POINTER(FgParams)

# name -> (restype, argtypes); every symbol include/formation_hip.h declares
SIGNATURES = {
    "fg_abi_version": (_I, []),
    "fg_last_error": (ctypes.c_char_p, []),
    "fg_launch_device": (_I, [_P, _P]),
    "fg_arena_create": (_I, [_I, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_uint64),
                             ctypes.POINTER(ctypes.c_uint32)]),
    "fg_arena_map": (_I, [_P, ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, ctypes.POINTER(_P)]),
    "fg_arena_unmap": (_I, [_P, _P]),
    "fg_arena_keep_window": (_I, [_P, _P, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(_P)]),
    "fg_arena_trim": (_I, [_P]),
    "fg_arena_retired_address_bytes": (ctypes.c_uint64, []),
    "fg_arena_destroy": (_I, [_P]),
    "fg_kernel_config": (_I, [_I, ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "fg_describe_launch": (_I, [_PP, ctypes.POINTER(FgScenario), _I, _I, _I, _I, _I, _I, ctypes.c_char_p, _I]),
    "fg_step_hd_bytes": (ctypes.c_int64, [_I]),
    "fg_step_hd": (_I, [_PP, _I, _I] + [_P] * 16),
    "fg_step_hd_plan": (_I, [_PP, _I, _I] + [_P] * 16 + [ctypes.POINTER(ctypes.c_void_p)]),
    "fg_plan_launch": (_I, [_P, ctypes.c_uint64]),
    "fg_plan_destroy": (_I, [_P]),
    "fg_physics_step": (_I, [_PP, _I, _I] + [_P] * 6),
    "fg_observe_hd": (_I, [_PP, _I, _I] + [_P] * 15),
    "fg_rollout_hd": (_I, [_PP, _I, _I, _I] + [_P] * 12 + [_I, _P]),
    "fg_reset_hd": (_I, [_PP, _I, _I] + [_P] * 9),
    "fg_reset_hd_mt": (_I, [_I, _I] + [_P] * 11),
    "fg_reset_hd_mt_done": (_I, [_I, _I, _I] + [_P] * 10 + [ctypes.c_int64, _P]),
    "fg_reset_scenario_mt": (_I, [ctypes.POINTER(FgScenario), _I, _I, _P, _I] + [_P] * 10),
    "fg_step_basic": (_I, [_PP, _I, _I, _I, _I] + [_P] * 13),
    "fg_step_scenario": (_I, [_PP, ctypes.POINTER(FgScenario), _I, _I, _I] + [_P] * 14),
    "fg_reset_scenario": (_I, [_PP, ctypes.POINTER(FgScenario), _I, _I] + [_P] * 10),
    "fg_rollout_scenario": (_I, [_PP, ctypes.POINTER(FgScenario), _I, _I, _I] + [_P] * 14 + [_I, _P]),
    "fg_decode_actions": (_I, [_I, ctypes.c_int64, _P, _P, _P]),
    "fg_update_comm": (_I, [_PP, _I, _I, _P, _P, _P]),
    "fg_update_comm_dim": (_I, [_PP, _I, _I, _I, _P, _P, _P]),
    "fg_policy_bfs": (_I, [_I, _I, _I, _P, ctypes.c_int64, _P, _P]),
    "fg_policy_bfs_state": (_I, [_I, _I, _I] + [_P] * 6),
    "fg_rollout_hd_policy": (_I, [_PP, _I, _I, _I, _I] + [_P] * 12 + [_I, _P]),
}

_lib = None


def build(force=False):
    """Compile the HIP library in-tree (hipcc --offload-arch=gfx950)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["bash", BUILD_SCRIPT])
    return LIB_PATH


def load():
    """Load the library, set prototypes, check the ABI version.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FormationHipError(
            FG_ERR_HIP, "%s not found - run gym-formation_amd/csrc/build.sh "
            "(there is no CPU fallback for the hot path)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.fg_abi_version() != ABI_VERSION:
        raise FormationHipError(FG_ERR_BAD_ARG, "ABI version mismatch: library %d, binding %d"
                                % (lib.fg_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(code):
    if code != FG_OK:
        raise FormationHipError(code, load().fg_last_error().decode("utf-8", "replace"))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream(device=None):
    """Raw handle of torch's current stream ON `device` (the device the env's tensors live on, which need
    not be the process's current device: the library switches to the stream's device for the launch)."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def current_stream_fast(device):
    """Raw handle of torch's current stream on `device` without building a Stream object
    (the per-step API path calls this on every step)."""
    import torch
    try:
        return torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())
    except AttributeError:                # private helper missing in this torch build
        return torch.cuda.current_stream(device).cuda_stream


def kernel_config(n):
    t, e, l = _I(), _I(), _I()
    check(load().fg_kernel_config(int(n), ctypes.byref(t), ctypes.byref(e), ctypes.byref(l)))
    return {"threads": t.value, "envs_per_wg": e.value, "lds_bytes": l.value}


def step_hd_bytes(n):
    return int(load().fg_step_hd_bytes(int(n)))
