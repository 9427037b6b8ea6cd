"""The one-env-per-lane kernels of the landmark scenarios (csrc/fg_scn_lane_kernel.hpp: every count a compile-time
constant) against the run-time-count kernel (fg::scn_kernel, `FgScenario.variant = 1`), through the C ABI: every output
and the whole state, bit for bit - single steps, K-step launches with episodes ending inside the launch, obs_every,
observation-only launches, batch sizes around the 64-env wave, contacts (crowded starts), World options.
The run-time-count kernel itself is held to the reference's fixtures and the oracle elsewhere
(tests/test_gpu_parity.py, test_gpu_fuzz_oracle.py).  Run with `pytest -m gpu`."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gym-formation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu

# (kind, N, L, M, num_obs, agent size, world_length, penalty): the reference's make_world defaults, and what
# make_env(name) makes of them with its own default num_agents = 3
SHAPES = [("basic", 3, 3, 0, 0, 0.1, 50, 1.0), ("partial", 5, 5, 0, 3, 0.04, 25, 1.0), ("range", 4, 4, 0, 0, 0.04, 25, 1.0),
          ("obstacle", 4, 4, 3, 0, 0.1, 50, 2.0), ("partial", 3, 5, 0, 3, 0.04, 25, 1.0), ("range", 3, 4, 0, 0, 0.04, 25, 1.0),
          ("obstacle", 3, 4, 3, 0, 0.1, 50, 2.0)]


def _setup(kind, N, L, M, num_obs, size, W, penalty, B, crowd, seed, auto_reset=True, **opts):
    from formation_gym import _native
    kinds = {"basic": _native.FG_SCN_BASIC, "partial": _native.FG_SCN_PARTIAL, "range": _native.FG_SCN_RANGE,
             "obstacle": _native.FG_SCN_OBSTACLE}
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    u = lambda *s: (torch.rand(s, generator=gen, device="cuda") * 2 - 1)
    st = dict(px=u(B, N) * crowd, py=u(B, N) * crowd, vx=u(B, N) * 0.3, vy=u(B, N) * 0.3, lm=u(B, L, 2).contiguous(),
              step=(torch.arange(B, device="cuda") % 7).to(torch.int32))
    if M:
        st["opos"] = (u(B, M, 2) * crowd).contiguous()                  # obstacles among the agents: agent-obstacle contacts
        st["ovel"] = torch.tensor([0.0, -1.0], device="cuda").expand(B, M, 2).contiguous()
    p = _native.FgParams(dt=0.1, damping=0.25, contact_force=100.0, contact_margin=1e-3, sensitivity=5.0, mass=1.0,
                         dist_min=2 * size, collide_thresh=2 * size, world_length=W, auto_reset=1 if auto_reset else 0,
                         seed=seed, rng_offset=5, env_index_base=3, **opts)
    sc = _native.FgScenario(kind=kinds[kind], num_landmarks=L, num_obstacles=M, num_obs=num_obs, obs_range=0.7,
                            obstacle_size=0.15, obstacle_vx=0.0, obstacle_vy=-1.0, obstacle_floor=-2.2, penalty=penalty)
    nbr = num_obs if kind == "partial" else N - 1
    D = 2 + (2 if kind == "basic" else 0) + 2 * L + 2 * M + 2 * nbr + 2 * (N - 1)
    return st, p, sc, D


def _rollout(st, p, sc, B, N, D, acts, obs_every, variant, near=False):
    from formation_gym import _native
    lib = _native.load()
    K = acts.shape[0]
    s = {k: v.clone() for k, v in st.items()}
    f = dict(dtype=torch.float32, device="cuda")
    out = dict(obs=torch.full((K // obs_every, B, N, D), 7.0, **f), rew=torch.full((K, B, N), 7.0, **f),
               indiv=torch.full((K, B, N), 7.0, **f), done=torch.full((K, B, N), 7, dtype=torch.uint8, device="cuda"))
    if near:
        out["near"] = torch.full((K, B, sc.num_landmarks), -1, dtype=torch.int32, device="cuda")
    sc.variant = variant
    _native.check(lib.fg_rollout_scenario(p, sc, B, N, K, s["px"].data_ptr(), s["py"].data_ptr(), s["vx"].data_ptr(),
                                          s["vy"].data_ptr(), acts.data_ptr(), s["lm"].data_ptr(), _native.ptr(s.get("opos")),
                                          _native.ptr(s.get("ovel")), s["step"].data_ptr(), out["obs"].data_ptr(),
                                          out["rew"].data_ptr(), out["indiv"].data_ptr(), out["done"].data_ptr(),
                                          _native.ptr(out.get("near")), obs_every, None))
    torch.cuda.synchronize()
    return s, out


def _same(a, b, what):
    for k in a:
        x, y = a[k], b[k]
        same = torch.equal(x, y) or bool(((x == y) | (x.isnan() & y.isnan())).all()) if x.is_floating_point() else torch.equal(x, y)
        assert same, "%s: %s differs (%d of %d elements)" % (what, k, int((x != y).sum()), x.numel())


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%s-%d-%d-%d" % s[:4])
@pytest.mark.parametrize("B", [1, 63, 64, 200, 4096])
def test_lane_kernel_equals_the_runtime_count_kernel(shape, B):
    kind, N, L, M = shape[:4]
    for crowd, K, obs_every in ((1.0, 1, 1), (0.25, 9, 1), (0.25, 8, 2)):
        st, p, sc, D = _setup(*shape, B=B, crowd=crowd, seed=11 * N + B)
        W = shape[6]
        st["step"] = torch.where(torch.arange(B, device="cuda") % 3 == 0, W - 4, 2).to(torch.int32)   # a third ends inside the launch
        gen = torch.Generator(device="cuda"); gen.manual_seed(B)
        acts = (torch.rand((K, B, N, 2), generator=gen, device="cuda") * 2 - 1).contiguous()
        s0, o0 = _rollout(st, p, sc, B, N, D, acts, obs_every, 0, near=kind == "basic")
        s1, o1 = _rollout(st, p, sc, B, N, D, acts, obs_every, 1, near=kind == "basic")
        what = "%s N=%d B=%d K=%d crowd=%.2f" % (kind, N, B, K, crowd)
        _same(o0, o1, what)
        _same(s0, s1, what)
        assert torch.isfinite(o0["obs"]).all() and not (o0["rew"] == 7.0).any() and not (o0["done"] == 7).any()
        if K > 4:
            assert (s0["step"] < W - 3).any() and bool(o0["done"].any()), "no episode ended inside the launch"
        if crowd < 1.0 and B >= 200:
            assert (o0["indiv"] != o0["indiv"][..., :1]).any() or M, "no collision penalty anywhere: the start is not crowded"


@pytest.mark.parametrize("shape,B,K,obs_every", [(SHAPES[0], 65536, 32, 1), (SHAPES[0], 393216 + 70, 6, 1), (SHAPES[1], 131072, 8, 1),
                                                 (SHAPES[2], 131072, 24, 2)],
                         ids=lambda v: "%s-%d" % v[:2] if isinstance(v, tuple) else str(v))
def test_wide_lane_kernel_equals_the_runtime_count_kernel(shape, B, K, obs_every):
    """The 256-env workgroups of the lane kernels (four producer waves; taken where they fill whole generations of 256 workgroups
    and the observation buffer is beyond the Infinity Cache: profiles/r05_lane_pw_ab.txt) against the run-time-count kernel, bit
    for bit, a ragged last workgroup and episodes ending inside the launch included."""
    from formation_gym import _native
    import ctypes
    kind, N, L, M = shape[:4]
    st, p, sc, D = _setup(*shape, B=B, crowd=0.3, seed=5 * N + K)
    assert (K // obs_every) * B * N * D * 4 > 400e6
    buf = ctypes.create_string_buffer(512)
    sc.variant = 0
    assert _native.load().fg_describe_launch(p, sc, B, N, K, 0, obs_every, 0, buf, len(buf)) == 0
    assert ("scn_lane_kernel<%d,%d,%d,%d," % (sc.kind, N, L, M)) in buf.value.decode() and ",4> grid" in buf.value.decode(), buf.value
    W = shape[6]
    st["step"] = torch.where(torch.arange(B, device="cuda") % 3 == 0, W - 4, 2).to(torch.int32)
    gen = torch.Generator(device="cuda"); gen.manual_seed(B)
    acts = (torch.rand((K, B, N, 2), generator=gen, device="cuda") * 2 - 1).contiguous()
    s0, o0 = _rollout(st, p, sc, B, N, D, acts, obs_every, 0)
    s1, o1 = _rollout(st, p, sc, B, N, D, acts, obs_every, 1)
    _same(o0, o1, "%s N=%d B=%d K=%d" % (kind, N, B, K))
    _same(s0, s1, "%s N=%d B=%d K=%d state" % (kind, N, B, K))
    assert bool(o0["done"].any()) and torch.isfinite(o0["obs"]).all()


@pytest.mark.parametrize("shape", SHAPES[:4], ids=lambda s: "%s-%d-%d-%d" % s[:4])
def test_lane_kernel_observation_only_and_world_options(shape):
    """do_physics = 0 (what env.reset() returns) and the World options (walls, speed clamp, accel, motor noise) go
    through the same helpers in both kernels."""
    from formation_gym import _native
    lib = _native.load()
    kind, N, L, M = shape[:4]
    B = 333
    st, p, sc, D = _setup(*shape, B=B, crowd=0.3, seed=5, auto_reset=False)
    outs = []
    for variant in (0, 1):
        sc.variant = variant
        f = dict(dtype=torch.float32, device="cuda")
        o = dict(obs=torch.zeros((B, N, D), **f), rew=torch.zeros((B, N), **f), indiv=torch.zeros((B, N), **f),
                 done=torch.zeros((B, N), dtype=torch.uint8, device="cuda"))
        s = {k: v.clone() for k, v in st.items()}
        _native.check(lib.fg_step_scenario(p, sc, B, N, 0, s["px"].data_ptr(), s["py"].data_ptr(), s["vx"].data_ptr(), s["vy"].data_ptr(),
                                           None, s["lm"].data_ptr(), _native.ptr(s.get("opos")), _native.ptr(s.get("ovel")),
                                           s["step"].data_ptr(), o["obs"].data_ptr(), o["rew"].data_ptr(), o["indiv"].data_ptr(),
                                           o["done"].data_ptr(), None))
        torch.cuda.synchronize()
        _same(s, st, "observe-only launch must not touch the state")
        outs.append(o)
    _same(outs[0], outs[1], "%s observe-only" % kind)
    st, p, sc, D = _setup(*shape, B=B, crowd=0.3, seed=6, accel=3.0, max_speed=0.4, u_noise=0.2, num_walls=2)
    p.walls[0] = _native.FgWall(vertical=0, axis_pos=0.2, end0=-0.5, end1=0.5, width=0.1, soft=0)
    p.walls[1] = _native.FgWall(vertical=1, axis_pos=-0.1, end0=-0.4, end1=0.6, width=0.05, soft=0)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    acts = (torch.rand((6, B, N, 2), generator=gen, device="cuda") * 2 - 1).contiguous()
    s0, o0 = _rollout(st, p, sc, B, N, D, acts, 1, 0)
    s1, o1 = _rollout(st, p, sc, B, N, D, acts, 1, 1)
    _same(o0, o1, "%s with World options" % kind)
    _same(s0, s1, "%s with World options" % kind)
    speed = (s0["vx"] ** 2 + s0["vy"] ** 2).sqrt()
    assert float(speed.max()) <= 0.4 * (1 + 1e-5)
