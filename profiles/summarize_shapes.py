#!/usr/bin/env python3
"""Condense profile_all_shapes.sh output (rocprofv3 kernel traces of bench.py at the four BASELINE per-GPU
shapes, one process per shape and launch mode) into one markdown table:
    python3 profiles/summarize_shapes.py gpurun_out/shapes_<tag> profiles/<tag>_all_shapes.md"""
import csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize import longest_run                          # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
lines = ["# rocprofv3 --kernel-trace --stats: all BASELINE per-GPU shapes, both launch modes (one process per shape and mode)",
         "",
         "Made by `profiles/profile_all_shapes.sh <tag>` (one gpurun call).  `algorithmic GB/s` = (24 N^2 + 53 N + 16) x envs x steps per launch / avg duration.",
         "A row is the TIMED series of its kernel (the longest run of back-to-back launches: bench.py's blocks), not every launch of the",
         "kernel: the placement probe's launches into candidate buffers that were not kept and the warm-up are left out (`all` = how many",
         "launches the kernel had in the process).  At 243 agents the same kernel serves single-step launches (pipelined over env batches)",
         "and 4-step rollout launches.", "",
         "| shape | mode | kernel | launches (all) | steps per launch | avg us | min us | max us | algorithmic GB/s | % of 8 TB/s | bench line of the same (profiled) process |",
         "|---|---|---|---|---|---|---|---|---|---|---|"]
notes = []


def key(p):
    n, mode = os.path.basename(p)[1:].split("_")
    return int(n), mode


for d in sorted(glob.glob(os.path.join(src, "n*_*")), key=key):
    n, mode = key(d)
    b = json.loads([l for l in open(os.path.join(d, "bench.json")) if l.startswith("{")][-1])
    B = b["config"]["envs_per_gpu"]; chunk = b["config"]["steps_per_launch"]
    bytes_step = (24 * n * n + 53 * n + 16) * B
    tr = glob.glob(os.path.join(d, "*", "*_kernel_trace.csv"))[0]
    groups = {}
    for r in csv.DictReader(open(tr)):
        if "fg::" in r["Kernel_Name"]:
            groups.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    name, spans = max(groups.items(), key=lambda kv: sum(e - s for s, e in kv[1]))      # the mode's own kernel
    short = name.replace("void ", "")
    if ">(" in short:
        short = short[:short.index(">(") + 1]                # drop the argument list
    ds = longest_run(spans)
    avg = sum(ds) / len(ds)
    g = bytes_step * chunk / (avg * 1e-6) / 1e9
    lines.append("| %d x %d | %s | `%s` | %d (%d) | %d | %.2f | %.2f | %.2f | %.0f | %.1f | %.3f us/step, %.1f %% |" % (
        n, B, mode, short, len(ds), len(spans), chunk, avg, min(ds), max(ds), g, g / 80.0, b["ms_per_step"] * 1e3,
        100 * b["roofline"]["frac"]))
    pl = b.get("placement") or {}
    txt = "; ".join("%s buffer %s MB, arena %s GB, kept '%s' %.4f ms vs as created %.4f" % (
        k, v.get("buffer_MB"), v.get("arena_GB"), v.get("kept"), v.get("kept_ms", 0), v.get("as_created_ms", 0))
        for k, v in pl.items() if v and v.get("probed"))
    notes.append("- %d agents, %s: %s" % (n, mode, txt or "buffer below the probe threshold, not placed"))
lines += ["", "Placement of the timed buffer in each process:", ""] + notes
open(dst, "w").write("\n".join(lines) + "\n")
print(open(dst).read())
