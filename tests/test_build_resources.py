"""Build gate (no GPU): no shipped kernel uses scratch memory.

Round 4 shipped 31 of 174 kernels with a private segment (the BASELINE config-5 kernel `rollout_kernel_wide<243,...>` among
them): spilled registers, a by-value array behind a pointer select, a hoisted table that did not fit.  The code object's metadata
says so without running anything, so this test reads it (`tests/isa_scan.py`: llvm-objdump --offloading + llvm-readelf --notes)
and fails on any kernel whose `private_segment_fixed_size` is not 0 or that spills vector registers.  SGPR spills into VGPR
lanes use no memory and are reported, not failed.  The fp64 build (tests only, never loaded by the product) is exempt."""
import glob
import os

import pytest

from formation_gym import _native

from . import isa_scan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gym-formation_amd", "csrc")


@pytest.fixture(scope="module")
def resources():
    sources = glob.glob(os.path.join(CSRC, "*.h*")) + [os.path.join(ROOT, "include", "formation_hip.h")]
    stale = (not os.path.exists(_native.LIB_PATH)
             or os.path.getmtime(_native.LIB_PATH) < max(os.path.getmtime(f) for f in sources))
    _native.build(force=stale)                          # the scan must describe the sources as they are
    return isa_scan.kernel_resources(_native.LIB_PATH)


def test_the_scan_sees_the_kernels(resources):
    names = [k["demangled"] for k in resources]
    assert len(resources) >= 150
    for needle in ("fg::step_kernel<27, 32, 256, 4", "fg::rollout_kernel<27, 32, 512, 512, 16, 10, 0, true>",
                   "fg::rollout_kernel_wide<243, 4, 4, 256, 0, false>", "fg::rollout_kernel_wide<243, 4, 4, 256, 0, true>",
                   "fg::scn_lane_kernel<", "fg::policy_bfs_kernel<3>"):
        assert any(needle in n for n in names), needle


def test_no_kernel_uses_scratch_memory(resources):
    bad = [(k["demangled"][:110], k["private_segment"], k["vgpr_spill"]) for k in resources
           if k["private_segment"] != 0 or k["vgpr_spill"] != 0]
    assert not bad, "kernels with a private segment / VGPR spills: %s" % bad


def test_register_budgets(resources):
    """The budgets the launch geometry counts on: <= 128 VGPRs where two 512-thread workgroups (or four waves per SIMD) share a
    CU, <= 168 for the 768- / 1024-thread pipelined kernels, <= 256 for the 512-thread wide ones."""
    for k in resources:
        n = k["demangled"]
        if "fg::rollout_kernel<" in n:
            assert k["vgpr"] <= 168, (n, k["vgpr"])
        if "fg::rollout_kernel_wide<81" in n or "fg::rollout_kernel_wide<125" in n:
            assert k["vgpr"] <= 128, (n, k["vgpr"])       # two workgroups per CU
        if "fg::step_kernel<0, 1024, 1024" in n:
            assert k["vgpr"] <= 128, (n, k["vgpr"])
