"""The dispatch of the library - which kernel instantiation and launch geometry a shape gets - as a committed snapshot.

`fg_describe_launch` walks the very dispatch code of fg_step_hd / fg_rollout_hd / fg_rollout_hd_policy / fg_rollout_scenario
without touching a device.  tests/golden/dispatch.json holds its answers over a grid of agent counts, batch sizes, steps per launch,
placed / ordinary buffers, open / closed loop and the landmark scenarios; a threshold that moves (on purpose or not) shows up as
a diff of that file in review instead of as a timing somebody may or may not re-measure ("a hand-tuned threshold forest ...
a 15 % regression of the headline passes every test").  Regenerate after an intended change:
    FG_UPDATE_DISPATCH_SNAPSHOT=1 python -m pytest tests/test_dispatch_snapshot.py
No GPU needed."""
import ctypes
import json
import os

import pytest

from formation_gym import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SNAPSHOT = os.path.join(ROOT, "tests", "golden", "dispatch.json")

AGENTS = [3, 4, 8, 9, 10, 16, 25, 27, 32, 64, 81, 100, 125, 243, 300]
BATCHES = [64, 256, 512, 1024, 2048, 2560, 3072, 4096, 5000, 8192, 10240, 14336, 16384, 32768, 65536]
STEPS = [0, 2, 20, 128]                                   # 0 = fg_step_hd, else fg_rollout_hd with K steps
PER = {3: 3, 4: 2, 8: 2, 9: 3, 16: 4, 25: 5, 27: 3, 32: 2, 64: 4, 81: 3, 125: 5, 243: 3}


@pytest.fixture(scope="module")
def lib():
    _native.build()
    return _native.load()


def _params(**kw):
    base = dict(dt=0.1, damping=0.25, contact_force=100.0, contact_margin=1e-3, sensitivity=5.0, mass=1.0, dist_min=0.03,
                collide_thresh=0.015, world_length=100, auto_reset=1)
    base.update(kw)
    return _native.FgParams(**base)


def _describe(lib, params, scenario, B, N, K, per=0, index_outputs=0):
    buf = ctypes.create_string_buffer(2048)
    rc = lib.fg_describe_launch(params, scenario, B, N, K, per, 1, index_outputs, buf, len(buf))
    return buf.value.decode().strip() if rc == 0 else "status %d" % rc


def _current(lib):
    out = {}
    for N in AGENTS:
        for B in BATCHES:
            if N * N * B > 243 * 243 * 8192:              # beyond what one GPU holds for a single step
                continue
            for K in STEPS:
                for placed in (0, 1):
                    if placed and K < 2:
                        continue
                    key = "hd N=%d B=%d K=%d placed=%d" % (N, B, K, placed)
                    out[key] = _describe(lib, _params(obs_placed=placed), None, B, N, K)
                    if N in PER and K >= 2:
                        out[key + " per=%d" % PER[N]] = _describe(lib, _params(obs_placed=placed), None, B, N, K, per=PER[N])
    # World options, index outputs, a padded env pitch: the run-time-N / OPTS instantiations
    for N, B in ((9, 4096), (27, 4096), (81, 2048), (243, 64), (1024, 4)):
        out["hd N=%d B=%d K=0 max_speed" % (N, B)] = _describe(lib, _params(max_speed=0.5), None, B, N, 0)
        out["hd N=%d B=%d K=20 max_speed" % (N, B)] = _describe(lib, _params(max_speed=0.5), None, B, N, 20)
        out["hd N=%d B=%d K=0 index outputs" % (N, B)] = _describe(lib, _params(), None, B, N, 0, index_outputs=1)
        pitch = -(-6 * N * N // 32) * 32
        out["hd N=%d B=%d K=20 pitch=%d" % (N, B, pitch)] = _describe(lib, _params(obs_env_pitch=pitch), None, B, N, 20)
    # the landmark scenarios at the reference's shapes and off them
    shapes = [("basic", _native.FG_SCN_BASIC, 3, 3, 0, 0), ("partial", _native.FG_SCN_PARTIAL, 5, 5, 0, 3),
              ("range", _native.FG_SCN_RANGE, 4, 4, 0, 0), ("obstacle", _native.FG_SCN_OBSTACLE, 4, 4, 3, 0),
              ("partial", _native.FG_SCN_PARTIAL, 3, 5, 0, 3), ("obstacle", _native.FG_SCN_OBSTACLE, 3, 4, 3, 0),
              ("basic", _native.FG_SCN_BASIC, 7, 7, 0, 0), ("obstacle", _native.FG_SCN_OBSTACLE, 70, 4, 3, 0)]
    for name, kind, N, L, M, num_obs in shapes:
        sc = _native.FgScenario(kind=kind, num_landmarks=L, num_obstacles=M, num_obs=num_obs, obs_range=0.5, obstacle_size=0.15,
                                obstacle_vx=0.0, obstacle_vy=-1.0, obstacle_floor=-2.2, penalty=1.0)
        for B in (100, 4096, 65536, 131072):
            for K in (1, 20, 128):
                out["%s N=%d L=%d M=%d B=%d K=%d" % (name, N, L, M, B, K)] = _describe(lib, _params(), sc, B, N, K)
    return out


def test_dispatch_matches_the_committed_snapshot(lib):
    now = _current(lib)
    if os.environ.get("FG_UPDATE_DISPATCH_SNAPSHOT"):
        with open(SNAPSHOT, "w") as fh:
            json.dump(now, fh, indent=0, sort_keys=True)
    with open(SNAPSHOT) as fh:
        want = json.load(fh)
    assert sorted(now) == sorted(want), "the grid changed: regenerate the snapshot"
    diff = {k: (want[k], now[k]) for k in now if now[k] != want[k]}
    assert not diff, "dispatch changed for %d shapes, e.g. %s" % (len(diff), list(diff.items())[:3])


def test_the_headline_shapes_take_the_kernels_the_profiles_name(lib):
    """The instantiations bench.py times (profiles/r05_*_rollout.md name them)."""
    d = lambda N, B, K, **kw: _describe(lib, _params(**kw), None, B, N, K)
    assert d(27, 4096, 20, obs_placed=1).startswith("rollout_kernel<27,32,512,512,16,10,0,1> grid 256")
    assert d(27, 4096, 20).startswith("rollout_kernel<27,32,512,256,16,10,0,1> grid 256")
    assert d(9, 4096, 128).startswith("rollout_kernel<9,16,256,512,16,64,0,0> grid 256")
    assert d(81, 2048, 20).startswith("rollout_kernel_wide<81,2,4,256,0,0> grid 512")
    assert d(243, 8192, 4).startswith("rollout_kernel_wide<243,4,4,256,0,0> grid 2048")
    assert d(243, 8192, 0).startswith("rollout_kernel_wide<243,4,4,256,0,1>")          # single step: pipelined over env batches
    assert d(27, 4096, 0).startswith("step_kernel<27,32,256,4,0,0,1> grid 1024")
    assert _describe(lib, _params(), None, 65536, 4, 20, per=2).startswith("hd_lane_kernel<4,2,1>")


def test_every_rule_names_a_measurement_that_is_in_the_repo():
    """The rule tables of the pipelined rollout kernels (csrc/formation_hip.hip: RollRule rows = batch range, buffer conditions,
    instantiation, the measurement behind the threshold): every profiles/... file a row cites exists, every table ends with a
    catch-all row, and rows are written one instantiation each (thresholds in one table with the profile file named)."""
    import glob
    import re
    src = open(os.path.join(ROOT, "gym-formation_amd", "csrc", "formation_hip.hip")).read()
    body = src[src.index("struct RollRule"):src.index("// The pipelined K-step kernels exist")]
    tables = re.findall(r"static constexpr RollRule rules\[\] = \{(.*?)\n    \};", body, flags=re.S)
    assert len(tables) == 4
    rows = 0
    for t in tables:
        entries = re.findall(r"\{\s*(\d+|B_ANY),\s*(\d+|B_ANY),\s*([^,]+),\s*roll_fn<([^>]*)>\(\),\s*((?:\"[^\"]*\"\s*)+)\}", t)
        assert entries, "no rows parsed"
        rows += len(entries)
        lo, hi, need, fn, why = entries[-1]
        assert (lo, hi, need.strip()) == ("0", "B_ANY", "0") and fn.split(",")[0].strip() == "true", "the last row must be a catch-all"
        assert t.count("roll_fn<") == len(entries), "a row the test could not parse"
        for lo, hi, need, fn, why in entries:
            assert len(why.strip()) > 10
    assert rows >= 20
    cited = {n.rstrip(".:,;") for n in re.findall(r"profiles/[A-Za-z0-9_./*-]+", body)}
    assert len(cited) >= 8
    for n in cited:
        assert glob.glob(os.path.join(ROOT, n) + "*"), "%s is cited by a dispatch rule but is not in the repo" % n
