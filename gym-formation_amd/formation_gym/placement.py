"""Placement probe for large observation buffers.

A K-step rollout streams its observations ([K, B, N, 6N] floats, 99 % of the bytes of the path) into one big
allocation.  On MI355X the rate of one and the same launch depends on WHICH allocation it writes to: fresh
allocations of the same size run it at 5.3 ... 6.3 TB/s, stable for the lifetime of the allocation (the driver's
physical page -> HBM channel placement; byte offsets inside an allocation do not matter, a physically contiguous
allocation is the worst case: profiles/r02_place/).  Nothing inside a kernel reaches that, so the host picks: allocate
a few candidates (all held at once, so that they land on different pages), time the caller's own launch on each, keep
the fastest and hand the others back to the driver.

Only buffers beyond the 256 MiB Infinity Cache are probed (smaller ones are absorbed by the cache) and the candidates
together never take more than a fraction of the device's free memory.
"""
import torch

MIN_PROBE_BYTES = 256 << 20


def probe_allocation(alloc, time_fn, nbytes, device, candidates=8, mem_fraction=0.6, reps=3, min_bytes=MIN_PROBE_BYTES,
                     good_enough=0.90):
    """Returns (buffer, report).  alloc() -> a fresh buffer (any object: a tensor, a dict of tensors); time_fn(buffer)
    enqueues ONE launch that streams into it on torch's current stream of `device`.  Up to `candidates` allocations are
    tried (all held until the end: a freed candidate's pages would come straight back), fewer when they would take more
    than `mem_fraction` of the free device memory; the search stops early once the rates have shown both modes, i.e. the
    best candidate takes <= `good_enough` x the time of the worst (the placements are bimodal, several levels 5-13 % apart:
    profiles/r03_placement.md).  report = {tried, ms per candidate, kept, kept_ms, worst_ms, ...}; with
    nbytes < min_bytes or candidates < 2 a single allocation is returned un-probed (report['tried'] == 1)."""
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
        time_fn(buf)                                            # first touch: page faults / TLB fill stay untimed
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record(stream)
        for r in range(reps):
            time_fn(buf)
            ev[r + 1].record(stream)
        stream.synchronize()
        t = sorted(ev[r].elapsed_time(ev[r + 1]) for r in range(reps))
        ms.append(t[len(t) // 2])
        if len(ms) >= 2 and min(ms) <= good_enough * max(ms):
            break
    best = min(range(len(ms)), key=lambda i: ms[i])
    keep = held[best]
    del held, buf
    torch.cuda.empty_cache()                                    # the losers go back to the driver, not to torch's pool
    return keep, {"tried": len(ms), "max_candidates": m, "ms": [round(x, 4) for x in ms], "kept": best, "probed": True,
                  "kept_ms": round(ms[best], 4), "worst_ms": round(max(ms), 4),
                  "worst_over_kept": round(max(ms) / ms[best], 4)}
