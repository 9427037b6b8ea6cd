#!/usr/bin/env python3
"""Is it safe to map different physical chunks at an address that another mapping has just left?  Mapping A is filled with
a pattern and unmapped, mapping B (other chunks) is filled with another pattern by MANY workgroups; then A is mapped again:
it must still hold its own pattern, and B must read back what was written.  A stale translation anywhere on the chip sends
part of B's accesses into A's chunks.

va_reuse_check.txt: the library at commit d98e924 handed reservations back (hipMemAddressFree), so B landed on A's address:
12 of 12 rounds corrupted, with or without an ordinary allocate / free cycle in between.
va_fresh_check.txt: the library as it is now retires every address range that has held a mapping, so B gets a fresh address.
   python profiles/r03_place/va_reuse_check.py [rounds] [plain|malloc]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
from formation_gym import _native, placement              # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
mode = sys.argv[2] if len(sys.argv) > 2 else "plain"
dev = torch.device("cuda:0")
W, CH = 8, 128 << 20
arena = placement.Arena(4 * W * CH, dev, CH)
nfl = W * CH // 4
bad = 0
same_va = 0
for r in range(rounds):
    A = list(range(0, W)) if r % 2 == 0 else list(range(2 * W, 3 * W))
    Bc = list(range(W, 2 * W)) if r % 2 == 0 else list(range(3 * W, 4 * W))
    va = arena.map(A)
    ta = arena.floats(va, nfl)
    ta.fill_(float(r + 1))
    torch.cuda.synchronize()
    del ta
    arena.unmap(va)
    if mode == "malloc":                                   # an ordinary allocate / free cycle between the two mappings
        x = torch.empty(64 << 20, dtype=torch.uint8, device=dev); del x; torch.cuda.empty_cache()
    vb = arena.map(Bc)
    same_va += int(vb == va)
    tb = arena.floats(vb, nfl)
    tb.fill_(-float(r + 1))
    torch.cuda.synchronize()
    okb = bool((tb == -float(r + 1)).all())
    del tb
    arena.unmap(vb)
    va2 = arena.map(A)
    ta = arena.floats(va2, nfl)
    oka = bool((ta == float(r + 1)).all())
    frac = float((ta != float(r + 1)).float().mean()) if not oka else 0.0
    del ta
    arena.unmap(va2)
    print("round %2d  A at %#x, B at %#x (%s), A again at %#x:  B reads back %s, A intact %s%s" % (
        r, va, vb, "same address" if vb == va else "other address", va2, okb, oka, "" if oka else "  (%.2f %% of A overwritten)" % (100 * frac)), flush=True)
    bad += int(not (oka and okb))
print("mode %s: %d of %d rounds corrupted; B landed on A's address in %d rounds" % (mode, bad, rounds, same_va))
arena.close()
