#!/usr/bin/env bash
# Counter passes around 20-step rollout launches of the landmark scenarios (where do the waves' cycles go?).
#   bash profiles/r03_scn_pmc.sh   (GPU box, repo root) -> gpurun_out/r03_scn_pmc.txt
R=$PWD
export TMPDIR=/tmp
OUT=$R/gpurun_out/scn_pmc; rm -rf $OUT; mkdir -p $OUT
PASSES=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
        "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY"
        "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FP64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE")
for cfg in "basic_formation_env 3 65536" "formation_hd_partial_env 5 65536" "formation_hd_obs_env 4 65536"; do
  set -- $cfg
  i=0
  for pass in "${PASSES[@]}"; do
    (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/$1_$i -- python3 $R/profiles/r03_scn_pmc.py $1 $2 $3 > $OUT/$1_$i.log 2>&1) || { echo "pass $i failed for $1"; tail -3 $OUT/$1_$i.log; }
    i=$((i + 1))
  done
done
python3 - $OUT <<'PY' | tee $R/gpurun_out/r03_scn_pmc.txt
import csv, glob, os, sys
out = sys.argv[1]
for sc in ("basic_formation_env", "formation_hd_partial_env", "formation_hd_obs_env"):
    vals = {}
    for f in glob.glob(os.path.join(out, sc + "_*", "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "scn_kernel" in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print("==", sc)
    for k in sorted(vals):
        v = sorted(vals[k]); print("  %-28s median %.4g  (launches %d)" % (k, v[len(v) // 2], len(v)))
PY
