#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (run_profile.sh) into a small JSON + markdown
summary that is committed under profiles/.

  python profiles/summarize.py gpurun_out/prof_<tag> profiles/<tag>

HBM traffic follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are
in KiB and collected in separate passes; on gfx950 FETCH_SIZE reports half of a
wide coalesced read stream, so the read side is doubled (this kernel's loads are
4/8 bytes per lane, for which the guide says the factor is uncalibrated - reads
are 3 % of the bytes here, so the uncertainty is < 2 % of the total)."""
import csv
import glob
import json
import os
import sys


def longest_run(spans, max_gap_us=200.0):
    """spans: (start_ns, end_ns) of one kernel's launches.  The longest series of back-to-back launches (gap to the
    previous one < max_gap_us) = bench.py's timed region: its blocks are enqueued without a host wait in between, while
    the placement probe's launches (3 per candidate buffer, formation_gym/placement.py), the warm-up and the other legs
    are shorter series separated by device synchronisations.  Returns the durations (us) of that series."""
    spans = sorted(spans)
    best, cur = [], []
    for i, (s, e) in enumerate(spans):
        if cur and (s - spans[i - 1][1]) / 1e3 > max_gap_us:
            cur = []
        cur.append((e - s) / 1e3)
        if len(cur) > len(best):
            best = cur
    return best


def main(src, dst):
    out = {"source": os.path.basename(src)}
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    kernels = []
    if stats:
        for r in csv.DictReader(open(stats[0])):
            if "fg::" in r["Name"]:
                kernels.append({"name": r["Name"], "calls": int(r["Calls"]),
                                "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                                "max_us": float(r["MaxNs"]) / 1e3, "pct": float(r["Percentage"])})
    out["kernel_stats"] = kernels
    trace = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
    if trace:
        rows = [r for r in csv.DictReader(open(trace[0])) if "fg::" in r["Kernel_Name"]]
        if rows:
            r = rows[-1]
            out["dispatch"] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                                                 "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size") if k in r}
            ts = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
            gaps = sorted(b[0] - a[1] for a, b in zip(ts, ts[1:]))
            if gaps:
                out["median_gap_between_launches_us"] = gaps[len(gaps) // 2] / 1e3
            if kernels:
                dom = max(kernels, key=lambda k: k["pct"])["name"]
                run = longest_run([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r["Kernel_Name"] == dom])
                if run:
                    # kernel_stats above averages EVERY launch of the kernel, the probe's launches into candidate buffers
                    # that were not kept included; this is the series bench.py times
                    out["timed_region"] = {"kernel": dom, "launches": len(run), "avg_us": round(sum(run) / len(run), 2),
                                           "min_us": round(min(run), 2), "max_us": round(max(run), 2),
                                           "how": "longest series of back-to-back launches in the kernel trace"}
    for kind, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = glob.glob(os.path.join(src, "pmc_" + kind, "*", "*_counter_collection.csv"))
        if not f:
            continue
        vals = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(f[0]))
                      if "fg::" in r["Kernel_Name"] and r["Counter_Name"] == name
                      and (not kernels or r["Kernel_Name"] == max(kernels, key=lambda k: k["pct"])["name"]))
        if vals:
            out[name + "_KiB_per_launch_median"] = vals[len(vals) // 2]
    # every other counter pass (profiles/r03_wide_pmc.sh: pmc_<name>/): median per launch of the dominant fg:: kernel
    dominant = max(kernels, key=lambda k: k["pct"])["name"] if kernels else None
    counters = {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d) or os.path.basename(d) in ("pmc_fetch", "pmc_write"):
            continue
        per = {}
        for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if dominant is None or r["Kernel_Name"] == dominant:
                    per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for name, vals in per.items():
            vals.sort()
            counters[name] = vals[len(vals) // 2]
    if counters:
        out["dominant_kernel"] = dominant
        out["counters_per_launch_median"] = counters
        if counters.get("TCC_EA0_WRREQ_sum"):
            out["full_64B_write_request_ratio"] = round(counters.get("TCC_EA0_WRREQ_64B_sum", 0.0) / counters["TCC_EA0_WRREQ_sum"], 5)
    if "FETCH_SIZE_KiB_per_launch_median" in out and "WRITE_SIZE_KiB_per_launch_median" in out:
        out["hbm_traffic_bytes_per_launch"] = int(1024 * (2 * out["FETCH_SIZE_KiB_per_launch_median"]
                                                          + out["WRITE_SIZE_KiB_per_launch_median"]))
    bench = os.path.join(src, "bench.json")
    if os.path.exists(bench):
        lines = [l for l in open(bench).read().strip().splitlines() if l.startswith("{")]
        if lines:
            out["bench"] = json.loads(lines[-1])
    with open(dst + ".json", "w") as f:
        json.dump(out, f, indent=1)
    with open(dst + ".md", "w") as f:
        f.write("# rocprofv3 summary: %s\n\n" % out["source"])
        wl = out.get("bench", {}).get("config", {}).get("workload", "27 agents x 4096 envs")
        f.write("command: `bash profiles/run_profile.sh <tag>` (or `profiles/r03_wide_pmc.sh`) = rocprofv3 --kernel-trace --stats, then\n"
                "--pmc passes (FETCH_SIZE, WRITE_SIZE, ... each in its own run), each around a short `python3 bench.py\n"
                "--no-cpu-baseline --no-extra` of the workload (%s), then an un-profiled bench run.\n\n" % wl)
        f.write("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
        for k in kernels:
            f.write("| `%s` | %d | %.2f | %.2f | %.2f | %.1f |\n" % (k["name"], k["calls"], k["avg_us"], k["min_us"], k["max_us"], k["pct"]))
        f.write("\n")
        if "timed_region" in out:
            t = out["timed_region"]
            f.write("The table averages every launch of a kernel, the placement probe's launches into candidate buffers included.\n"
                    "bench.py's timed region alone (%s): %d launches of the dominant kernel, avg %.2f us, min %.2f, max %.2f.\n\n"
                    % (t["how"], t["launches"], t["avg_us"], t["min_us"], t["max_us"]))
        for k, v in out.items():
            if k not in ("kernel_stats", "bench", "source", "timed_region"):
                f.write("- %s: %s\n" % (k, v))
        if "bench" in out:
            b = out["bench"]
            f.write("\nun-profiled bench line of the same build: value %.4g %s, ms/step %.5f, roofline %s\n"
                    % (b["value"], b["unit"], b["ms_per_step"], json.dumps(b["roofline"])))
    print(open(dst + ".md").read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
