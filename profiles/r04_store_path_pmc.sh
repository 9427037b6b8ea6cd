#!/usr/bin/env bash
# Where do the stores of a rollout launch wait?  TA / TCP / translation counters around the rollout kernels of
# 9 x 4096 x 128 (writer-bound), 9 x 8192 x 64 and 27 x 4096 x 20 (the headline, stream-bound), placed buffers.
#   bash profiles/r04_store_path_pmc.sh   (GPU box, repo root) -> gpurun_out/r04_store_path_pmc.txt
R=$PWD
export TMPDIR=/tmp
OUT=$R/gpurun_out/store_path_pmc; rm -rf $OUT; mkdir -p $OUT
PASSES=("TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
        "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum"
        "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
        "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TOTAL_WRITE_sum"
        "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum")
for shape in 9:4096:128 9:8192:64 27:4096:20; do
  i=0
  for pass in "${PASSES[@]}"; do
    (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/${shape//:/_}_$i -- python3 $R/profiles/r03_rollout_ab.py $shape > $OUT/${shape//:/_}_$i.log 2>&1) || { echo "pass $i failed for $shape"; tail -3 $OUT/${shape//:/_}_$i.log; }
    i=$((i + 1))
  done
done
python3 - $OUT <<'PY' | tee $R/gpurun_out/r04_store_path_pmc.txt
import csv, glob, os, sys
out = sys.argv[1]
print("# rocprofv3 --pmc around the rollout kernel of three shapes (median over the launches of the kernel in one process per pass)")
for shape in ("9_4096_128", "9_8192_64", "27_4096_20"):
    vals = {}
    for f in glob.glob(os.path.join(out, shape + "_*", "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "rollout_kernel" in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print("== %s" % shape.replace("_", " x "))
    for k in sorted(vals):
        v = sorted(vals[k])
        print("  %-48s median %.5g  (launches %d)" % (k, v[len(v) // 2], len(v)))
PY
