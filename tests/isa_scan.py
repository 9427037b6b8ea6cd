"""Kernel resource usage of a built HIP shared library, read from the code object's metadata notes (no GPU needed).

`kernel_resources(path)` -> [{name, demangled, private_segment, vgpr, sgpr, vgpr_spill, sgpr_spill}, ...] for the gfx950 code
object embedded in `path`.  Used by tests/test_build_resources.py (every shipped kernel runs without scratch memory) and as a
command:  python3 tests/isa_scan.py [lib.so]   prints every kernel that has a private segment or spills."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def kernel_resources(lib_path, arch="gfx950"):
    tmp = tempfile.mkdtemp(prefix="fg_isa_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)
        # llvm-objdump --offloading writes every bundle entry next to its input: <input>.<n>.<triple>
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", local], check=True, cwd=tmp,
                       stdout=subprocess.DEVNULL)
        objs = [f for f in os.listdir(tmp) if f.endswith(arch)]
        if not objs:
            raise RuntimeError("no %s code object in %s" % (arch, lib_path))
        out = []
        for obj in objs:
            notes = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", os.path.join(tmp, obj)], check=True,
                                   capture_output=True, text=True).stdout
            for block in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
                def field(key, default=0):
                    m = re.search(r"\.%s:\s+(\d+)" % key, block)
                    return int(m.group(1)) if m else default
                name = re.search(r"\.name:\s+(\S+)", block).group(1)
                out.append({"name": name, "private_segment": field("private_segment_fixed_size"), "vgpr": field("vgpr_count"),
                            "sgpr": field("sgpr_count"), "vgpr_spill": field("vgpr_spill_count"),
                            "sgpr_spill": field("sgpr_spill_count")})
        names = "\n".join(k["name"] for k in out)
        filt = shutil.which("c++filt") or os.path.join(LLVM_BIN, "llvm-cxxfilt")
        try:
            dem = subprocess.run([filt], input=names, capture_output=True, text=True, check=True).stdout.split("\n")
        except (OSError, subprocess.CalledProcessError):
            dem = names.split("\n")
        for k, d in zip(out, dem):
            k["demangled"] = d
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "gym-formation_amd", "lib", "libformation_hip.so")
    ks = kernel_resources(lib)
    bad = [k for k in ks if k["private_segment"] or k["vgpr_spill"]]
    for k in sorted(bad, key=lambda k: k["demangled"]):
        print("private %4d B  vgpr %3d  vgpr spills %3d  sgpr spills %3d  %s"
              % (k["private_segment"], k["vgpr"], k["vgpr_spill"], k["sgpr_spill"], k["demangled"][:130]))
    spilled = [k for k in ks if k["sgpr_spill"] and not (k["private_segment"] or k["vgpr_spill"])]
    print("%d kernels, %d with a private segment or VGPR spills, %d more with SGPR spills (to VGPR lanes) only"
          % (len(ks), len(bad), len(spilled)))
    for k in sorted(spilled, key=lambda k: -k["sgpr_spill"])[:40]:
        print("   sgpr spills %3d  vgpr %3d  %s" % (k["sgpr_spill"], k["vgpr"], k["demangled"][:130]))
