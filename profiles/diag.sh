#!/usr/bin/env bash
# PMC passes over profiles/diag.py workloads; prints mean per launch of each counter.
export TMPDIR=/tmp
export R=$PWD
OUT=$R/gpurun_out/diag; mkdir -p $OUT
PASSES=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
        "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_WRITE_sum"
        "TCC_REQ_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_NORMAL_WRITEBACK_sum"
        "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE"
        "TA_BUSY_avr TA_FLAT_WRITE_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
        "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_NC_WRITE_REQ_sum")
cd /tmp
i=0
for w in "$@"; do
  tag=$(echo $w | tr ' ' '_')
  p=0
  for pass in "${PASSES[@]}"; do
    rocprofv3 --pmc $pass --output-format csv -d $OUT/${tag}_p$p -- python3 $R/profiles/diag.py $w > $OUT/${tag}_p$p.log 2>&1
    p=$((p+1))
  done
  echo "== $w"
  FG_TAG=$tag python3 - <<'PY'
import csv, glob, os, collections
tag = os.environ["FG_TAG"]
agg = collections.defaultdict(list)
dur = []
for f in glob.glob("/root/repo/gpurun_out/diag/%s_p*/*/*_counter_collection.csv" % tag) + glob.glob(os.environ.get("R", ".") + "/gpurun_out/diag/%s_p*/*/*_counter_collection.csv" % tag):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "step_kernel" in k or "FillFunctor<float>" in k:
            if int(r["Grid_Size"]) < 100000: continue
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k][5:]
    print("  %-40s %14.1f" % (k, sum(v) / max(1, len(v))))
PY
done
