#!/usr/bin/env python3
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "fg::" in r["Kernel_Name"] and "reset" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["physics_only", "step_without_obs", "full_step"]
n = len(rows) // 3
for k, name in enumerate(names):
    grp = rows[k * n:(k + 1) * n][10:]
    d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp)
    gaps = sorted(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(grp, grp[1:]))
    print("%-18s kernel us: median %.2f min %.2f | gap to next launch median %.2f us" %
          (name, d[len(d) // 2] / 1e3, d[0] / 1e3, gaps[len(gaps) // 2] / 1e3))
