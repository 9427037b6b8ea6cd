#!/usr/bin/env python3
"""Condense profile_all_shapes.sh output (rocprofv3 kernel traces of bench.py at the four BASELINE per-GPU
shapes) into one markdown table:  python3 profiles/summarize_shapes.py gpurun_out/shapes_<tag> profiles/<tag>_all_shapes.md"""
import csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize import longest_run                          # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
lines = ["# rocprofv3 --kernel-trace --stats: all BASELINE per-GPU shapes, both launch modes in one process",
         "",
         "Made by `profiles/profile_all_shapes.sh <tag>` (one gpurun call).  `algorithmic GB/s` = (24 N^2 + 53 N + 16) x envs x steps per launch / avg duration.",
         "At N = 243 the same kernel serves single-step launches (pipelined over env batches) and 4-step rollout launches; the two groups are",
         "split by duration from the kernel trace.  A row is the TIMED series of its group (the longest run of back-to-back launches:",
         "bench.py's blocks), not every launch of the kernel: the placement probe's launches into candidate buffers that were not kept,",
         "the warm-up and the counter legs are left out (`all` = how many launches the group had in the process).", "",
         "| shape | kernel | launches (all) | steps per launch | avg us | min us | max us | algorithmic GB/s | % of 8 TB/s |", "|---|---|---|---|---|---|---|---|---|"]
for d in sorted(glob.glob(os.path.join(src, "n*")), key=lambda p: int(os.path.basename(p)[1:])):
    n = int(os.path.basename(d)[1:])
    b = json.loads(open(os.path.join(d, "bench.json")).readline())
    B = b["config"]["envs_per_gpu"]; chunk = b["config"]["steps_per_launch"]
    bytes_step = (24 * n * n + 53 * n + 16) * B
    tr = glob.glob(os.path.join(d, "*", "*_kernel_trace.csv"))[0]
    groups = {}
    for r in csv.DictReader(open(tr)):
        if "fg::" not in r["Kernel_Name"]:
            continue
        groups.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    for name, spans in groups.items():
        if len(spans) < 5:
            continue
        durs = [(e - s) / 1e3 for s, e in spans]
        short = name.replace("void ", "")
        if ">(" in short:
            short = short[:short.index(">(") + 1]          # drop the argument list
        parts = [(spans, chunk if "rollout_kernel" in name else 1)]
        if "rollout_kernel_wide<243" in name:
            thr = 2.0 * min(durs)
            parts = [([x for x in spans if (x[1] - x[0]) / 1e3 < thr], 1), ([x for x in spans if (x[1] - x[0]) / 1e3 >= thr], chunk)]
        for sp, k in parts:
            if len(sp) < 3:
                continue
            ds = longest_run(sp)
            avg = sum(ds) / len(ds)
            g = bytes_step * k / (avg * 1e-6) / 1e9
            lines.append("| %d x %d | `%s` | %s | %d | %.2f | %.2f | %.2f | %.0f | %.1f |" % (n, B, short, "%d (%d)" % (len(ds), len(sp)), k, avg, min(ds), max(ds), g, g / 80.0))
    lines.append("| %d x %d | bench line of the same (profiled) process: %s mode %.3f us/step, other mode %.3f us/step | | | | | | | |" % (
        n, B, b["config"]["mode"], b["ms_per_step"] * 1e3, b["other_mode"]["ms_per_step"] * 1e3))
lines += ["", "Under rocprofv3 the memory a first arena hands back (the step buffer's probe) is not yet free when the second arena of the",
          "process is made, so the rollout buffer's arena is smaller there (arena GB below; 206 GB in a process of its own) and its chunks less",
          "spread: the 81-agent rollout row is ~4 % slower here than in `<tag>_81x2048_rollout.md`, which profiles the rollout mode alone.", ""]
for d in sorted(glob.glob(os.path.join(src, "n*")), key=lambda p: int(os.path.basename(p)[1:])):
    b = json.loads(open(os.path.join(d, "bench.json")).readline())
    pl = b.get("placement") or {}
    lines.append("- %s agents: " % os.path.basename(d)[1:] + ("; ".join(
        "%s buffer %s MB, arena %s GB (free before: %s GB), kept '%s' %.4f ms vs as created %.4f" % (
            k, v.get("buffer_MB"), v.get("arena_GB"), v.get("free_GB_before"), v.get("kept"), v.get("kept_ms", 0), v.get("as_created_ms", 0))
        for k, v in pl.items() if v and v.get("probed")) or "buffers below the probe threshold, not placed"))
open(dst, "w").write("\n".join(lines) + "\n")
print(open(dst).read())
