#!/usr/bin/env python3
"""Condense profile_all_shapes.sh output (rocprofv3 kernel traces of bench.py at the four BASELINE per-GPU
shapes) into one markdown table:  python3 profiles/summarize_shapes.py gpurun_out/shapes_<tag> profiles/<tag>_all_shapes.md"""
import csv, glob, json, os, sys

src, dst = sys.argv[1], sys.argv[2]
lines = ["# rocprofv3 --kernel-trace --stats: all BASELINE per-GPU shapes, both launch modes in one process",
         "",
         "Made by `profiles/profile_all_shapes.sh <tag>` (one gpurun call).  `algorithmic GB/s` = (24 N^2 + 53 N + 16) x envs x steps per launch / avg duration.",
         "At N = 243 the same kernel serves single-step launches (pipelined over env batches) and 4-step rollout launches; the two groups are",
         "split by duration from the kernel trace.", "",
         "| shape | kernel | launches | steps per launch | avg us | min us | max us | algorithmic GB/s | % of 8 TB/s |", "|---|---|---|---|---|---|---|---|---|"]
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
        groups.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for name, durs in groups.items():
        if len(durs) < 5:
            continue
        short = name.replace("void ", "")
        if ">(" in short:
            short = short[:short.index(">(") + 1]          # drop the argument list
        parts = [(durs, chunk if "rollout_kernel" in name else 1)]
        if "rollout_kernel_wide<243" in name:
            thr = 2.0 * min(durs)
            parts = [([x for x in durs if x < thr], 1), ([x for x in durs if x >= thr], chunk)]
        for ds, k in parts:
            if len(ds) < 3:
                continue
            avg = sum(ds) / len(ds)
            g = bytes_step * k / (avg * 1e-6) / 1e9
            lines.append("| %d x %d | `%s` | %d | %d | %.2f | %.2f | %.2f | %.0f | %.1f |" % (n, B, short, len(ds), k, avg, min(ds), max(ds), g, g / 80.0))
    lines.append("| %d x %d | bench line of the same (profiled) process: %s mode %.3f us/step, other mode %.3f us/step | | | | | | | |" % (
        n, B, b["config"]["mode"], b["ms_per_step"] * 1e3, b["other_mode"]["ms_per_step"] * 1e3))
open(dst, "w").write("\n".join(lines) + "\n")
print(open(dst).read())
