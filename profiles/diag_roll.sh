#!/usr/bin/env bash
# PMC passes on the default rollout bench (short), each under its own timeout.
export TMPDIR=/tmp
export R=$PWD; OUT=$R/gpurun_out/diagroll; rm -rf $OUT; mkdir -p $OUT
PASSES=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
        "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_WRITE_sum"
        "TCC_REQ_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_NORMAL_WRITEBACK_sum"
        "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
        "TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum")
cd /tmp
p=0
for pass in "${PASSES[@]}"; do
  timeout 240 rocprofv3 --pmc $pass --output-format csv -d $OUT/p$p -- python3 $R/bench.py --steps 120 --warmup 60 --no-cpu-baseline --no-extra $BENCH_ARGS > $OUT/p$p.log 2>&1
  p=$((p+1))
done
python3 - <<'PY'
import csv, glob, collections, os
agg = collections.defaultdict(list)
for f in glob.glob(os.environ.get("R", "/root/repo") + "/gpurun_out/diagroll/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fg::" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = sorted(agg[k]); v = v[len(v)//2:]           # full 20-step launches are the larger half
    print("  %-44s %16.1f  (n=%d)" % (k, sum(v) / max(1, len(v)), len(v)))
PY
