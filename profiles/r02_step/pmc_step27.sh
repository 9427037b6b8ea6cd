#!/usr/bin/env bash
# SQ counters of the single-step kernel at 27 x 4096 (one gpurun call, repo root): three --pmc passes around
# `python3 profiles/r02_generic_n.py 27:4096` (counter passes only: no tracing domains alongside).
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc_step27
mkdir -p $O
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/p1 -- python3 $R/profiles/r02_generic_n.py 27:4096 > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/p2 -- python3 $R/profiles/r02_generic_n.py 27:4096 > $O/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d $O/p3 -- python3 $R/profiles/r02_generic_n.py 27:4096 > $O/p3.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmc_step27/%s/**/*counter_collection.csv" % p, recursive=True):
        for row in csv.DictReader(open(f)):
            if "step_kernel<27" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        v = sorted(v)
        print("%-24s launches %5d  median per launch %14.0f" % (k, len(v), v[len(v) // 2]))
PY
