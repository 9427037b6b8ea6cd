"""CPU-side check of INTEGRATION.md's reference-side binding block: it parses, mirrors the header's FgParams
(ctypes layout == the product binding's) and names only exported symbols.  The GPU run of the same block is
tests/test_gpu_integration_stub.py."""
import ast
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_stub_parses_and_mirrors_the_header():
    from formation_gym import _native
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    src = [b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "class HipHotPath" in b]
    assert len(src) == 1
    tree = ast.parse(src[0])
    # evaluate only the two ctypes.Structure classes (no library load, no torch)
    ns = {"ctypes": ctypes}
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name in ("FgWall", "FgParams"):
            exec(compile(ast.Module([node], []), "INTEGRATION.md", "exec"), ns)
    assert ctypes.sizeof(ns["FgParams"]) == ctypes.sizeof(_native.FgParams)
    assert [(n, t) for n, t in ns["FgParams"]._fields_ if n != "walls"] == \
           [(n, t) for n, t in _native.FgParams._fields_ if n != "walls"]
    lib = _native.load() if os.path.exists(_native.LIB_PATH) else None
    for sym in set(re.findall(r"lib\.(fg_[a-z_0-9]+)", src[0])):
        assert sym in _native.SIGNATURES, sym
        if lib is not None:
            assert hasattr(lib, sym)
    assert "ABI version %d" % _native.ABI_VERSION in src[0]
