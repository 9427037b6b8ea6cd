"""Placement of large observation buffers in HBM.

A K-step rollout streams its observations ([K, B, N, 6N] floats, 99 % of the bytes of the path) into one big buffer.  On
MI355X the rate of one and the same launch depends on WHERE that buffer lies: windows of one large allocation run it at
5.2 ... 6.1 TB/s, in a pattern a few GB wide that follows the physical memory behind the addresses (profiles/r03_place/:
the same virtual addresses are fast in one process and slow in the next; single 1 GiB chunks all run alike, so it is the
combination a multi-GB buffer lands on; byte offsets below ~50 MB change nothing; a physically contiguous allocation is
the worst case, profiles/r02_place/).  Nothing inside a kernel reaches that, so the host places the buffer:

  `probe_arena`       address space backed by separately created physical chunks (`fg_arena_*`: HIP virtual memory
                      management); the caller's own launch is timed on windows of the arena at a stride of a quarter
                      window, then around the best one at a finer stride; the best window's chunks are kept where they
                      are, all others go back to the driver.  Nothing is wasted once the probe is over.
  `probe_allocation`  the fallback where the arena cannot be made: a few whole allocations held side by side, the
                      fastest kept (coarser: an allocation is one sample of the pattern).

Only buffers beyond the 256 MiB Infinity Cache are placed (smaller ones are absorbed by the cache), and the probe never
takes more than a fraction of the device's free memory.
"""
import ctypes

import torch

from . import _native

MIN_PROBE_BYTES = 256 << 20


def _time_launch(time_fn, buf, stream, reps):
    time_fn(buf)                                                # first touch: page faults / TLB fill stay untimed
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record(stream)
    for r in range(reps):
        time_fn(buf)
        ev[r + 1].record(stream)
    stream.synchronize()
    t = sorted(ev[r].elapsed_time(ev[r + 1]) for r in range(reps))
    return t[len(t) // 2]


def probe_allocation(alloc, time_fn, nbytes, device, candidates=8, mem_fraction=0.6, reps=3, min_bytes=MIN_PROBE_BYTES,
                     good_enough=0.90, min_tried=4):
    """Returns (buffer, report).  alloc() -> a fresh buffer (any object: a tensor, a dict of tensors); time_fn(buffer)
    enqueues ONE launch that streams into it on torch's current stream of `device`.  Up to `candidates` allocations are
    tried (all held until the end: a freed candidate's pages would come straight back), fewer when they would take more
    than `mem_fraction` of the free device memory; after `min_tried` candidates the search stops once the rates have
    shown both ends, i.e. the best candidate takes <= `good_enough` x the time of the worst.  report = {tried, ms per
    candidate, kept, kept_ms, worst_ms, ...}; with nbytes < min_bytes or candidates < 2 a single allocation is returned
    un-probed (report['tried'] == 1)."""
    device = torch.device(device)
    free = torch.cuda.mem_get_info(device)[0] if device.type == "cuda" else 0
    m = int(candidates)
    if nbytes < min_bytes:
        m = 1
    else:
        m = max(1, min(m, int(mem_fraction * free // max(1, nbytes))))
    if m < 2:
        return alloc(), {"tried": 1, "ms": [], "kept": 0, "probed": False}
    held, ms = [], []
    stream = torch.cuda.current_stream(device)
    for _ in range(m):
        buf = alloc()
        held.append(buf)
        ms.append(_time_launch(time_fn, buf, stream, reps))
        if len(ms) >= min_tried and min(ms) <= good_enough * max(ms):
            break
    best = min(range(len(ms)), key=lambda i: ms[i])
    keep = held[best]
    del held, buf
    torch.cuda.empty_cache()                                    # the losers go back to the driver, not to torch's pool
    return keep, {"method": "allocations", "tried": len(ms), "max_candidates": m, "ms": [round(x, 4) for x in ms], "kept": best,
                  "probed": True, "kept_ms": round(ms[best], 4), "worst_ms": round(max(ms), 4),
                  "worst_over_kept": round(max(ms) / ms[best], 4)}


class _Raw(object):
    """A device address range as something torch.as_tensor understands."""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (int(nfloats),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class Arena(object):
    """Address space backed by separately created physical chunks (C ABI `fg_arena_create / _keep / _destroy`).  The
    tensors made by `floats()` are views of it: keep the Arena alive as long as they are in use."""

    def __init__(self, nbytes, device, chunk_bytes=0):
        self.device = torch.device(device)
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        handle, base, chunk = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
        _native.check(_native.load().fg_arena_create(int(index), int(nbytes), int(chunk_bytes), ctypes.byref(handle),
                                                     ctypes.byref(base), ctypes.byref(chunk)))
        self._handle, self.base, self.chunk = handle, int(base.value), int(chunk.value)
        self.chunks = -(-int(nbytes) // self.chunk)

    def floats(self, byte_offset, nfloats):
        return torch.as_tensor(_Raw(self.base + int(byte_offset), nfloats), device=self.device)

    def keep(self, byte_offset, nbytes):
        _native.check(_native.load().fg_arena_keep(self._handle, int(byte_offset), int(nbytes)))

    def close(self):
        if self._handle is not None:
            h, self._handle = self._handle, None
            torch.cuda.synchronize(self.device)
            _native.load().fg_arena_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:                 # noqa: BLE001 - interpreter shutdown
            pass


def probe_arena(nfloats, time_fn, device, factor=8.0, mem_fraction=0.7, reps=3, min_bytes=MIN_PROBE_BYTES):
    """Returns (flat float32 tensor of `nfloats`, report, arena) - the tensor is a view of the arena, which the caller
    keeps alive - or None when the buffer is too small to matter or the arena cannot be made (the caller then falls back
    to `probe_allocation`).  time_fn(flat_tensor) enqueues ONE launch that streams into the candidate window."""
    device = torch.device(device)
    nbytes = int(nfloats) * 4
    if device.type != "cuda" or nbytes < min_bytes:
        return None
    free = torch.cuda.mem_get_info(device)[0]
    chunk = (1 << 30) if nbytes >= (4 << 30) else (256 << 20) if nbytes >= (1 << 30) else (64 << 20)
    total = int(min(factor * nbytes, mem_fraction * free))
    if total < nbytes + 2 * chunk:
        return None
    try:
        arena = Arena(total, device, chunk)
    except _native.FormationHipError:
        return None
    chunk = arena.chunk
    W = -(-nbytes // chunk)                                      # chunks per window
    last = arena.chunks - W
    stream = torch.cuda.current_stream(device)
    seen = {}

    def rate(k):
        if k not in seen:
            seen[k] = _time_launch(time_fn, arena.floats(k * chunk, nfloats), stream, reps)
        return seen[k]

    coarse = max(1, W // 4)
    for k in range(0, last + 1, coarse):
        rate(k)
    best = min(seen, key=seen.get)
    fine = max(1, coarse // 4)
    for k in range(max(0, best - coarse + fine), min(last, best + coarse - fine) + 1, fine):   # around the best window
        rate(k)
    best = min(seen, key=seen.get)
    ms = [seen[k] for k in sorted(seen)]
    stream.synchronize()
    arena.keep(best * chunk, nbytes)
    flat = arena.floats(best * chunk, nfloats)
    report = {"method": "arena windows", "probed": True, "tried": len(seen), "arena_GB": round(arena.chunks * chunk / 1e9, 1),
              "chunk_MiB": chunk >> 20, "window_chunks": W, "kept_window": best, "kept_ms": round(seen[best], 4),
              "worst_ms": round(max(ms), 4), "median_ms": round(sorted(ms)[len(ms) // 2], 4),
              "worst_over_kept": round(max(ms) / seen[best], 4)}
    return flat, report, arena
