"""Placement of large observation buffers in HBM.

A K-step rollout streams its observations ([K, B, N, 6N] floats, 99 % of the bytes of the path) into one big buffer.  On
MI355X the rate of one and the same launch depends on WHERE that buffer lies in physical memory (profiles/r03_place/):
windows of one large allocation run it at 5.0 ... 6.0 TB/s; the same virtual addresses are fast in one process and slow
in the next; single 1 GiB chunks all run alike, so it is the combination a multi-GB buffer lands on - and the rule behind
it is SPREAD: a buffer whose chunks are scattered over 160 GB of the device's memory runs at 6.1-6.5 TB/s where
neighbouring chunks give 5.0-5.4 (a 46 GB buffer with its own chunks merely shuffled: 6.8 instead of 6.1; a physically
contiguous allocation is the worst case, 2.0-2.6 TB/s, profiles/r02_place/).  Nothing inside a kernel reaches that, so
the host places the buffer:

  `probe_arena`       address space backed by separately created physical chunks (`fg_arena_*`: HIP virtual memory
                      management); candidate buffers are composed of chunks SPREAD over the whole arena (one per
                      stratum, shuffled; regions taken round-robin) and timed with the caller's own launch; the best
                      candidate's chunks are kept, all others go back to the driver.  Nothing is wasted afterwards.
                      The arena need not be large (round 4, profiles/r04_place/arena_size.txt): what pays is the TIMED
                      choice among compositions, not the distance - 6 x the buffer (8.6 GB for the 1.4 GB headline
                      buffer) gives what 206 GB gave on most boxes; where an arena gains nothing (some boxes' first
                      10 GB run every composition alike and slow) a second one, four times as large, is probed;
                      placing chunks by rule without timing does not work (the driver decides where a chunk lies:
                      spread_rule*.txt).
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


class _LaunchError(Exception):
    """Wraps an exception raised by the caller's own launch (`time_fn`) inside a probe, so that it is not mistaken for an
    arena / mapping failure (which falls back to whole allocations with a warning)."""


_live_arenas = []          # weak references to the arenas of this process: `is_placed` answers from them


def is_placed(address):
    """Does this device address lie in a buffer composed by `probe_arena` (chunks spread over the device's memory)?
    The scenario shells pass the answer to the library as `FgParams.obs_placed`."""
    address = int(address)
    alive = []
    hit = False
    for ref in _live_arenas:
        arena = ref()
        if arena is None or arena._handle is None:
            continue
        alive.append(ref)
        lo, hi = arena.kept_range
        if lo <= address < hi:
            hit = True
    _live_arenas[:] = alive
    return hit


class _Raw(object):
    """A device address range as something torch.as_tensor understands.  torch keeps this object alive for as long as
    the tensor's storage lives (any view of it included), and the object keeps the arena: a placed buffer's memory is
    released when its last tensor goes away, not when some owner remembers to."""

    def __init__(self, ptr, nfloats, owner=None):
        self.__cuda_array_interface__ = {"shape": (int(nfloats),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
        self.owner = owner


class Arena(object):
    """A set of separately created physical chunks (C ABI `fg_arena_*`) from which buffers are composed: `map(indices)`
    gives a device address range made of exactly those chunks; a chunk is mapped at one address at a time.  Tensors made
    by `floats()` are views of a mapping: keep the Arena alive as long as they are in use."""

    def __init__(self, nbytes, device, chunk_bytes=0):
        self.device = torch.device(device)
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        handle, chunk, count = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint32()
        _native.check(_native.load().fg_arena_create(int(index), int(nbytes), int(chunk_bytes), ctypes.byref(handle),
                                                     ctypes.byref(chunk), ctypes.byref(count)))
        self._handle, self.chunk, self.chunks = handle, int(chunk.value), int(count.value)
        self.kept_range = (0, 0)          # address range of the spread buffer kept in the end (see is_placed)
        import weakref
        _live_arenas.append(weakref.ref(self))

    def floats(self, address, nfloats):
        """A flat float32 tensor over `nfloats` floats at a device address inside one of the arena's mappings."""
        return torch.as_tensor(_Raw(int(address), nfloats, self), device=self.device)

    def map(self, chunk_index):
        """The given chunks (any order) mapped at fresh contiguous addresses; returns the base address."""
        arr = (ctypes.c_uint32 * len(chunk_index))(*[int(c) for c in chunk_index])
        base = ctypes.c_void_p()
        _native.check(_native.load().fg_arena_map(self._handle, arr, len(chunk_index), ctypes.byref(base)))
        return int(base.value)

    def unmap(self, base):
        _native.check(_native.load().fg_arena_unmap(self._handle, ctypes.c_void_p(int(base))))

    def keep_window(self, base, first, count):
        """Shrinks the mapping at `base` to `count` chunks from position `first` (the rest is unmapped); returns its address."""
        new_base = ctypes.c_void_p()
        _native.check(_native.load().fg_arena_keep_window(self._handle, ctypes.c_void_p(int(base)), int(first), int(count),
                                                          ctypes.byref(new_base)))
        return int(new_base.value)

    def trim(self):
        """Hand every chunk that is not mapped right now back to the driver."""
        _native.check(_native.load().fg_arena_trim(self._handle))

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


def default_arena_bytes(nbytes):
    """Arena size the probe starts with: 6 x the buffer up to 48 GiB, at least 1.5 x the buffer
    (profiles/r04_place/: the 1.4 GB headline buffer runs at 11.8-12.1 us/step from an 8.6 GB arena, 11.7-12.0 from 206 GB,
    12.2-13.0 from 4.3 GB; the 6.4 GB buffer of 81 x 2048 x 20 at 50.3-52.6 from 10 GB on four boxes and at 62.5 - nothing
    gained - on a fifth, at 50.5-50.9 from 39, 69 and 206 GB: a multi-GB buffer needs its 6 x as well)."""
    return int(max(1.5 * nbytes, min(6 * nbytes, 48 << 30)))


def arena_geometry(nbytes, free_bytes, mem_fraction=0.5, max_arena_bytes=None):
    """(arena bytes, chunk bytes) for a buffer of `nbytes` on a device with `free_bytes` free, or None when there is no
    room to choose from.  8 ... 16 chunks per buffer, 32 MiB ... 1 GiB each (the chunk size itself does not matter:
    profiles/r03_place/spread_*); the arena is `default_arena_bytes` (or `max_arena_bytes`), never more than `mem_fraction`
    of the free memory (a buffer larger than half of that: 1.5 x the buffer, at most 0.9 of the free memory); at most ~2000
    chunks per arena (driver calls, page tables)."""
    chunk = 32 << 20
    while chunk < (1 << 30) and nbytes // chunk > 16:
        chunk <<= 1
    if max_arena_bytes is None:
        max_arena_bytes = default_arena_bytes(nbytes)
    total = int(min(max_arena_bytes, mem_fraction * free_bytes))
    if total < 2 * nbytes:
        total = int(min(0.9 * free_bytes, 1.5 * nbytes))          # a huge buffer: at least some room to shuffle in
    while chunk < (1 << 30) and total // chunk > 2048:
        chunk <<= 1
    if total < nbytes + 2 * chunk:
        return None
    return total, chunk


def probe_arena(nfloats, time_fn, device, trials=8, mem_fraction=0.5, max_arena_bytes=None, reps=3,
                min_bytes=MIN_PROBE_BYTES, seed=0, budget_s=1.0, escalate=True):
    """Returns (flat float32 tensor of `nfloats`, report, arena) - the tensor lives in the arena, which the caller keeps
    alive - or None when the buffer is too small to matter or the arena cannot be made (the caller then falls back to
    `probe_allocation`).  time_fn(flat_tensor) enqueues ONE launch that streams into the candidate buffer.

    Candidates: the arena's first chunks in the order they were created (what a plain allocation gives) and selections of
    chunks SPREAD over the whole arena - one chunk per stratum, strata in shuffled or golden-stride order
    (profiles/r03_place/: the wider a buffer's chunks are spread over the device's memory, the faster the launch;
    neighbouring memory is the slow case).  At least `trials` selections are timed, up to eight times as many when
    `budget_s` affords them (a 0.25 ms launch can afford many, and its selections differ by 20 %; a 7 ms launch's differ
    by 1 %).  One candidate is mapped at a time (a chunk never has two addresses), the three fastest are mapped and timed
    once more, the winner is mapped for good and every other chunk goes back to the driver.  An arena whose winner is not
    8 % faster than its own first chunks is followed by one four times as large (`next_arena_bytes`, at most two such steps,
    report["stages"]; the better of the stages' winners is kept, the other arenas closed); a caller that passes `max_arena_bytes` or escalate=False gets exactly one arena.
    An error of the launch itself (`time_fn` raising) propagates: only arena / mapping failures fall back."""
    import math
    import random
    import time
    t_start = time.perf_counter()
    device = torch.device(device)
    nbytes = int(nfloats) * 4
    if device.type != "cuda" or nbytes < min_bytes:
        return None
    free = torch.cuda.mem_get_info(device)[0]
    geometry = arena_geometry(nbytes, free, mem_fraction, max_arena_bytes)
    if geometry is None:
        return None
    stages = []
    previous = None
    held = None        # the winner of an earlier stage, kept (its arena holds only the buffer by then) while a larger arena is tried
    while True:
        placed = _probe_stage(geometry, nfloats, time_fn, device, trials, reps, seed + len(stages), budget_s, free, t_start,
                              stage_index=len(stages))
        if placed is None:
            if held is not None:                                 # the larger arena could not be made or used: what we have
                held[1]["stages"] = stages + [{"arena_GB": round(geometry[0] / 1e9, 1), "failed": True}]
                return held
            if previous is None:
                return None
            # (no winner held: cannot happen since an escalation keeps one) the size that worked, once more, and no further stage
            placed = _probe_stage(previous, nfloats, time_fn, device, trials, reps, seed, budget_s, free, t_start)
            if placed is None:
                return None
            placed[1]["stages"] = stages + [{"arena_GB": placed[1]["arena_GB"], "kept_ms": placed[1]["kept_ms"],
                                             "as_created_ms": placed[1]["as_created_ms"], "after_failed_stage": True}]
            return placed
        stages.append({"arena_GB": placed[1]["arena_GB"], "kept_ms": placed[1]["kept_ms"], "as_created_ms": placed[1]["as_created_ms"]})
        if held is not None:
            # the better of the two winners stays, the other arena goes back to the driver (a larger arena is a different
            # sample of the device's memory, not a superset: its best composition can be the slower one)
            keep, drop = (held, placed) if held[1]["kept_ms"] <= placed[1]["kept_ms"] else (placed, held)
            drop_arena = drop[2]
            held = placed = drop = None
            drop_arena.close()
            placed = keep
        flat, report, arena = placed
        report["stages"] = stages
        # Some boxes hand out a first ~10 GB in which EVERY composition of a multi-GB buffer runs alike and slow
        # (profiles/r04_place/README.md: 81 x 2048 x 20 at 62.5 us/step from a 10 GB arena, nothing gained over the plain
        # allocation, where other boxes reach 50.5): when the arena offered too little, look at four times as much memory, twice
        # at most, within `mem_fraction` of what is free.  A caller that names the arena size gets that size.
        bigger = next_arena_bytes(geometry[0], nbytes, free, mem_fraction)
        # ... unless the launch does not care where its buffer lies: when the arena's first chunks and the median spread candidate ran within 2.5 % of the
        # best (a launch bound by its dependent chain, e.g. 9 x 4096 x 128: 0.2015 ... 0.2074 ms), a larger arena has nothing
        # to offer either - round 4 took it through 6 / 24 / 98 GB for nothing, 0.6 TB of address space
        spread = report.get("spread_ms_min_median_max") or [report["kept_ms"]] * 3
        best_ms = min(report["kept_ms"], report["as_created_ms"], spread[0])
        insensitive = max(spread[1], report["as_created_ms"]) <= ESCALATE_INSENSITIVE * best_ms     # the MEDIAN: one slow outlier is noise
        if (max_arena_bytes is not None or not escalate or nbytes < ESCALATE_MIN_BYTES or len(stages) >= 3 or bigger is None
                or insensitive or stages[0]["arena_GB"] * 1e9 >= ESCALATE_MAX_FIRST_ARENA or report["kept_ms"] <= ESCALATE_BELOW_GAIN * report["as_created_ms"]):
            return placed
        held = placed
        del flat, report, arena, placed
        previous = geometry
        geometry = arena_geometry(nbytes, free, mem_fraction, bigger)
        if geometry is None:                                     # cannot happen (bigger > what worked), but never loop on it
            return held


# A probe whose winner is not at least 8 % faster than the arena's first chunks has learnt too little from this arena: where
# placement matters at all (the median candidate is not within 2.5 % of the best, see ESCALATE_INSENSITIVE) a normal first arena
# offers 12-23 % (27 x 4096 x 20: 0.2335-0.24 ms against 0.27-0.28 as created; 9 x 4096 x 128: 0.190 against 0.246); on some boxes
# its best composition is only 5-7 % ahead (round 5: 0.2555 against 0.2705, the bench at 0.77 of peak instead of 0.83) and the
# good memory lies outside the first 6-9 GB (round 4's fresh box: 0.73 from 8.7 and 34.5 GB, 0.83 from 138 GB).  (3 % until late
# round 5.)
ESCALATE_BELOW_GAIN = 0.92
# Every probed buffer gets the second look: on one fresh box the first 8.7 GB ran ALL 65 spread compositions of the 1.4 GB
# headline buffer slower than its first chunks (0.283-0.290 ms against 0.267; a normal arena: 0.236-0.243), 0.73 instead of
# 0.82 of peak.  The price is paid by launches that placement cannot help: 9 x 4096 x 128, bound by its dependent chain, goes
# through 6 / 24 / 98 GB for 0.2339 / 0.2317 / 0.2334 ms, ~1.2 s once per buffer shape.
ESCALATE_MIN_BYTES = 0
# ... but not for launches that do not care where their buffer lies (median candidate within 2.5 % of the best), and not
# when the first arena was a large sample of the device's memory already (a 6.4 GB buffer starts with 39 GB; the escalation
# exists for the 1-2 GB buffers whose first 6-9 GB can be all alike).  Four times such an arena would retire hundreds of GB
# of address space per probe.
ESCALATE_INSENSITIVE = 1.025
ESCALATE_MAX_FIRST_ARENA = 32 << 30


def next_arena_bytes(total, nbytes, free_bytes, mem_fraction=0.5):
    """Size of the next, larger arena to probe for a buffer of `nbytes` after an arena of `total` bytes gained nothing:
    4 x as large, at most `mem_fraction` of the free memory; None when that is not at least twice what was tried."""
    bigger = int(min(4 * total, mem_fraction * free_bytes))
    return bigger if bigger >= 2 * total and bigger >= 2 * nbytes else None


def _probe_stage(geometry, nfloats, time_fn, device, trials, reps, seed, budget_s, free, t_start, stage_index=0):
    """One arena of probe_arena: (flat, report, arena) or None.  stage_index > 0: a follow-up arena of an escalating probe."""
    import math
    import random
    import time
    nbytes = int(nfloats) * 4
    total, chunk = geometry
    try:
        arena = Arena(total, device, chunk)
    except _native.FormationHipError:                            # no arena (driver, address space): whole allocations
        return None

    def probe():
        chunk, n = arena.chunk, arena.chunks
        W = -(-nbytes // chunk)                                      # chunks per buffer
        stream = torch.cuda.current_stream(device)
        rnd = random.Random(seed)

        def timed(idx, r, placed):                                   # map, time, unmap: one candidate at a time
            addr = arena.map(idx)
            arena.kept_range = (addr, addr + W * chunk) if placed else (0, 0)   # a spread candidate is timed as what it will be
            try:
                flat = arena.floats(addr, nfloats)
                try:
                    return _time_launch(time_fn, flat, stream, r)
                except Exception as exc:                             # noqa: BLE001 - the CALLER's launch failed: not a placement problem
                    raise _LaunchError(exc) from exc
            finally:
                stream.synchronize()
                arena.kept_range = (0, 0)
                arena.unmap(addr)

        def time_window(base, start, r):                             # W chunks of a whole-arena mapping, from position `start`
            addr = base + start * chunk
            arena.kept_range = (addr, addr + W * chunk)                  # a spread candidate is timed as what it will be
            try:
                flat = arena.floats(addr, nfloats)
                try:
                    return _time_launch(time_fn, flat, stream, r)
                except Exception as exc:                             # noqa: BLE001 - the CALLER's launch failed: not a placement problem
                    raise _LaunchError(exc) from exc
            finally:
                flat = None
                stream.synchronize()
                arena.kept_range = (0, 0)

        def arena_orders(t):
            # What makes a selection fast (profiles/r03_place/selection_rules_*.txt): chunks that are CONSECUTIVE in the
            # buffer must lie far apart in memory (>= 8-20 GB) - two clusters at the ends of the arena written one after the
            # other run like neighbours, the same clusters taken alternately run like the best spread.  Three orders of ALL the
            # arena's chunks in which EVERY window of W consecutive positions has that property; the candidates are windows
            # of one mapping per order (round 4 made one mapping per candidate and retired 65 x the buffer of address space per
            # probe instead of 3 x the arena):
            R = 3 + (t % 3)                                           # R regions of the arena taken round-robin (kept most often)
            per = n // R
            order = [(k % R) * per + k // R for k in range(R * per)] + list(range(R * per, n))
            yield "regions round-robin (%d)" % R, order
            g = max(1, int(round(n * 0.6180339887)))
            while math.gcd(g, n) != 1:
                g += 1
            off = rnd.randrange(n)
            yield "spread, golden stride", [(off + k * g) % n for k in range(n)]   # neighbours of the buffer 0.62 arenas apart
            order = list(range(n))
            rnd.shuffle(order)
            yield "spread, shuffled", order

        first = list(range(W))
        cands = [("as created", first)]
        last_base, where = None, {}                                  # the mapping kept after the loop, candidate -> window start in it
        t_warm = time.perf_counter()                             # bring the clocks up first: the early candidates of a cold
        addr = arena.map(first)                                  # probe measured 5-8 % slow
        try:
            flat0 = arena.floats(addr, nfloats)
            try:
                while time.perf_counter() - t_warm < 0.05:
                    _time_launch(time_fn, flat0, stream, 2)
                ms = [_time_launch(time_fn, flat0, stream, reps)]    # "as created": the arena's first chunks, on the same mapping
            except Exception as exc:                             # noqa: BLE001 - see timed()
                raise _LaunchError(exc) from exc
        finally:
            flat0 = None
            stream.synchronize()
            arena.unmap(addr)
        # a short launch is timed more often: the median of 3 launches of 0.24 ms is good to ~1.5 %, which is what separates
        # the best selections (one of six fresh processes kept a selection 2.4 % slower than the others' for it); every
        # candidate gets >= 3 ms of timed launches, at most 15 of them
        reps_c = int(min(15, max(reps, math.ceil(3.0 / max(ms[0], 1e-3)))))
        # how many selections the budget affords (a 0.25 ms launch many, and its selections differ by 20 %; a 7 ms launch
        # few, and its selections differ by 1 %)
        count = int(max(int(trials), min(8 * int(trials), budget_s / max(1e-6, (reps_c + 1) * ms[0] * 1e-3 + 4e-3))))
        # one mapping of the whole arena per order: three orders for buffers up to 2 GB, two up to 4 GB, one above; a follow-up
        # arena (escalation) is looked at through ONE order - it is four and sixteen times as large
        orders = list(arena_orders(seed))[:1 if stage_index > 0 else 3 if nbytes < (2 << 30) else 2 if nbytes < (4 << 30) else 1]
        per_order = -(-count // len(orders))
        refine = ms[0] < 2.0                                     # short launches: the three fastest once more (below)
        for o_i, (kind, order) in enumerate(orders):
            last = n - W                                         # last window start
            starts = sorted({int(round(j * last / max(1, per_order - 1))) for j in range(per_order)}) if last > 0 else [0]
            base = arena.map(order)                              # every chunk of the arena, once per order
            keep = False
            try:
                for st in starts:
                    cands.append((kind, order[st:st + W]))
                    ms.append(time_window(base, st, reps_c))
                    if o_i == len(orders) - 1:
                        where[len(ms) - 1] = st
                # the winner of a long launch lies in this (the last) mapping: keep its window where it is
                keep = (not refine) and o_i == len(orders) - 1 and min(range(len(ms)), key=lambda i: ms[i]) in where
            finally:
                if keep:
                    last_base = base
                else:
                    arena.unmap(base)
                    if o_i == len(orders) - 1:
                        where = {}
        if refine:
            # the three fastest once more, with more repetitions - and "as created" with them: the kept buffer is never one
            # that measured slower than what a plain allocation would have given
            finalists = sorted(set(sorted(range(len(ms)), key=lambda i: ms[i])[:3]) | {0})
            final = {i: timed(cands[i][1], 2 * reps_c + 1, i > 0) for i in finalists}
            for i, v in final.items():
                ms[i] = v
            best = min(final, key=final.get)
        else:                                                        # a launch of milliseconds is timed well enough the first time
            best = min(range(len(ms)), key=lambda i: ms[i])
        if last_base is not None and best in where:
            addr = arena.keep_window(last_base, where[best], W)      # the winner stays mapped, the rest of that mapping goes
        else:
            addr = arena.map(cands[best][1])                         # the winner, for good ...
        arena.trim()                                                 # ... and every other chunk back to the driver
        if best > 0:                                                 # every candidate but "as created" is a spread buffer
            arena.kept_range = (addr, addr + W * chunk)
        flat = arena.floats(addr, nfloats)
        spread = sorted(ms[1:])
        report = {"method": "arena: chunks spread over the device memory", "probed": True, "tried": len(ms),
                  "arena_GB": round(n * chunk / 1e9, 1), "free_GB_before": round(free / 1e9, 1), "chunk_MiB": chunk >> 20, "buffer_chunks": W, "kept": cands[best][0],
                  "kept_ms": round(ms[best], 4), "as_created_ms": round(ms[0], 4),
                  "spread_ms_min_median_max": [round(spread[0], 4), round(spread[len(spread) // 2], 4), round(spread[-1], 4)] if spread else [],
                  "worst_ms": round(max(ms), 4), "worst_over_kept": round(max(ms) / ms[best], 4),
                  "launches_per_candidate": reps_c, "probe_seconds": round(time.perf_counter() - t_start, 2),
                  # address ranges that held a candidate are retired, never reused (stale translations: include/formation_hip.h)
                  "retired_address_space_GB": round(_native.load().fg_arena_retired_address_bytes() / 1e9, 1)}
        return flat, report, arena

    try:
        return probe()
    except _LaunchError as exc:                                  # a genuine kernel / launch failure inside time_fn: the caller's
        arena.close()                                            # to see, not to be reported as "arena placement failed"
        raise exc.args[0]
    except Exception as exc:                                     # noqa: BLE001 - a mapping could not be made (address space,
        # driver), or the mapped range could not be used as a tensor on this device: placement is an optimisation, the
        # caller falls back to whole allocations
        import warnings
        warnings.warn("formation_gym.placement: arena placement failed (%s: %s); using whole allocations"
                      % (type(exc).__name__, exc), RuntimeWarning)
        arena.close()
        return None
