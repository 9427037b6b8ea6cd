#!/usr/bin/env python3
"""Joins a counter pass of profiles/r05_place_channels.sh to its candidate log: per candidate composition the kernel's duration
(profiler timestamps, launch-side HIP events) and every collected counter - total, and how it spreads over the 128 TCC instances
(16 channels x 8 XCDs): max / mean over instances, over channels (summed over XCDs), over XCDs (summed over channels).
   python3 profiles/r05_place_channels_join.py gpurun_out/r05_place/<pass> [--instances]"""
import glob
import json
import os
import sys

import numpy as np

d = sys.argv[1]
show_inst = "--instances" in sys.argv
log = json.load(open(os.path.join(d, "cands.json")))
res = json.load(open(glob.glob(os.path.join(d, "prof", "**", "*_results.json"), recursive=True)[0]))["rocprofiler-sdk-tool"][0]
names = {c["id"]["handle"]: c["name"] for c in res["counters"]}
ksym = {k["kernel_id"]: k.get("formatted_kernel_name", k.get("kernel_name", "")) for k in res["kernel_symbols"]}
rows = []
for rec in res["callback_records"]["counter_collection"]:
    dd = rec["dispatch_data"]
    if "rollout_kernel" not in ksym.get(dd["dispatch_info"]["kernel_id"], ""):
        continue
    vals = {}
    for r in rec["records"]:
        vals.setdefault(names[r["counter_id"]["handle"]], []).append(r["value"])
    rows.append((dd["dispatch_info"]["dispatch_id"], dd["end_timestamp"] - dd["start_timestamp"], vals))
rows.sort()
need = sum(c["launches"] for c in log["candidates"])
assert len(rows) == need, (len(rows), need)
counters = sorted(rows[0][2])
print("shape %d x %d x %d, buffer of %d chunks of %d MiB from an arena of %d; %d timed launches per candidate"
      % (log["N"], log["B"], log["K"], log["buffer_chunks"], log["chunk_MiB"], log["arena_chunks"], log["reps"]))
hdr = "%-36s %9s %9s" % ("candidate", "event us", "kernel us")
for c in counters:
    hdr += " | %-44s" % (c[:28] + ": total, max/mean inst, chan, xcd")
print(hdr)
pos = 0
table = []
for cand in log["candidates"]:
    mine = rows[pos + 1:pos + cand["launches"]]             # the first launch into a mapping pays its page-table walks
    pos += cand["launches"]
    dur = float(np.median([m[1] for m in mine])) / 1e3
    line = "%-36s %9.1f %9.1f" % (cand["label"][:36], cand["event_ms"] * 1e3, dur)
    stats = {"label": cand["label"], "event_us": cand["event_ms"] * 1e3, "kernel_us": dur}
    for c in counters:
        a = np.mean([m[2][c] for m in mine], axis=0)         # [instances], record order = XCD-major as the tool lists them
        tot = a.sum()
        if a.size == 128:
            g = a.reshape(8, 16)
            # which axis is the channel and which the XCD is the tool's listing order; both spreads are printed
            s0, s1 = g.sum(axis=0), g.sum(axis=1)
            line += " | %12.4g %8.3f %8.3f %8.3f    " % (tot, a.max() / max(a.mean(), 1e-9), s0.max() / max(s0.mean(), 1e-9),
                                                        s1.max() / max(s1.mean(), 1e-9))
            stats[c] = (tot, a)
        else:
            line += " | %12.4g %8s %8s %8s    " % (tot, "-", "-", "-")
            stats[c] = (tot, a)
    table.append(stats)
    print(line)
body = [t for t in table if not t["label"].startswith("warm-up")]
k = np.array([t["kernel_us"] for t in body])
print("\ncorrelation of the kernel's duration with each counter over the %d candidates (Pearson r); fastest %.1f us, slowest %.1f us"
      % (len(body), k.min(), k.max()))
for c in counters:
    tot = np.array([t[c][0] for t in body])
    mx = np.array([t[c][1].max() / max(t[c][1].mean(), 1e-9) for t in body])
    r_tot = np.corrcoef(k, tot)[0, 1] if tot.std() > 0 else float("nan")
    r_mx = np.corrcoef(k, mx)[0, 1] if mx.std() > 0 else float("nan")
    print("  %-36s total r = %+.2f   (fastest %.4g, slowest %.4g)   max/mean-instance r = %+.2f" %
          (c, r_tot, tot[k.argmin()], tot[k.argmax()], r_mx))
if show_inst:
    fast, slow = body[int(k.argmin())], body[int(k.argmax())]
    for c in counters:
        if fast[c][1].size != 128:
            continue
        print("\n%s per instance: fastest candidate (%s, %.1f us) | slowest (%s, %.1f us); rows = first listed dimension (8), columns = second (16)"
              % (c, fast["label"], fast["kernel_us"], slow["label"], slow["kernel_us"]))
        gf, gs = fast[c][1].reshape(8, 16), slow[c][1].reshape(8, 16)
        for x in range(8):
            print("  " + " ".join("%8.0f" % v for v in gf[x]) + "   |   " + " ".join("%8.0f" % v for v in gs[x]))
