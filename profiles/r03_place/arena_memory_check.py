#!/usr/bin/env python3
"""Does the device memory of an arena come back when its chunks are released (fg_arena_trim / fg_arena_destroy) although
the address ranges they were mapped at stay reserved (retired, include/formation_hip.h)?  Free device memory after each step.
   python profiles/r03_place/arena_memory_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-formation_amd")]
import torch                                              # noqa: E402
from formation_gym import placement                       # noqa: E402

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)


def free_gb(what):
    torch.cuda.synchronize()
    f, t = torch.cuda.mem_get_info(dev)
    print("%-58s free %.1f GB of %.1f" % (what, f / 1e9, t / 1e9), flush=True)
    return f


CH = 256 << 20
f0 = free_gb("start")
for r in range(2):
    arena = placement.Arena(256 * CH, dev, CH)            # 64 GiB
    free_gb("arena %d created (64 GiB, every chunk mapped once, unmapped)" % r)
    va = arena.map([3, 77, 200, 131])
    t = arena.floats(va, 4 * CH // 4); t.fill_(1.0); torch.cuda.synchronize(); del t
    arena.trim()
    free_gb("arena %d trimmed to the 4 mapped chunks (1 GiB)" % r)
    arena.unmap(va)
    arena.close()
    f1 = free_gb("arena %d destroyed" % r)
print("memory not returned: %.2f GB" % ((f0 - f1) / 1e9))
