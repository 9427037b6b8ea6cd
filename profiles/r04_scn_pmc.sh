#!/usr/bin/env bash
# Counter passes around K-step rollout launches of the landmark scenarios at 65536 envs (> 1 GB of observations per launch):
# the one-env-per-lane kernels (variant 0) and the run-time-count kernel (variant 1) - where do the waves' cycles go, how many
# scalar / branch instructions per vector instruction?
#   bash profiles/r04_scn_pmc.sh   (GPU box, repo root) -> gpurun_out/r04_scn_pmc.txt
R=$PWD
export TMPDIR=/tmp
OUT=$R/gpurun_out/scn_pmc_r04; rm -rf $OUT; mkdir -p $OUT
PASSES=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
        "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY"
        "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INSTS_VALU_TRANS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
        "WRITE_SIZE" "FETCH_SIZE")
for cfg in "basic_formation_env 3" "formation_hd_partial_env 5" "formation_hd_partial_range_env 4" "formation_hd_obs_env 4"; do
  set -- $cfg
  for variant in 0 1; do
    i=0
    for pass in "${PASSES[@]}"; do
      (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/$1_v${variant}_$i -- python3 $R/profiles/r04_scn_pmc.py $1 $2 65536 $variant > $OUT/$1_v${variant}_$i.log 2>&1) || { echo "pass $i failed for $1 variant $variant"; tail -3 $OUT/$1_v${variant}_$i.log; }
      i=$((i + 1))
    done
  done
done
python3 - $OUT <<'PY' | tee $R/gpurun_out/r04_scn_pmc.txt
import csv, glob, os, sys
out = sys.argv[1]
print("# rocprofv3 --pmc around K-step rollout launches of the landmark scenarios, 65536 envs, > 1 GB of observations per launch")
print("# (medians over the launches of one process per pass; WRITE_SIZE / FETCH_SIZE in KiB per launch)")
for sc in ("basic_formation_env", "formation_hd_partial_env", "formation_hd_partial_range_env", "formation_hd_obs_env"):
    for variant, kernel in ((0, "scn_lane_kernel"), (1, "scn_kernel<")):
        vals = {}
        for f in glob.glob(os.path.join(out, "%s_v%d_*" % (sc, variant), "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if kernel in r["Kernel_Name"]:
                    vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        print("== %s, %s" % (sc, "one env per lane (scn_lane_kernel)" if variant == 0 else "run-time counts (scn_kernel)"))
        med = {k: sorted(v)[len(v) // 2] for k, v in vals.items()}
        for k in sorted(med):
            print("  %-28s median %.4g  (launches %d)" % (k, med[k], len(vals[k])))
        if "SQ_INSTS_VALU" in med and "SQ_INSTS_SALU" in med:
            print("  -> SALU / VALU = %.3f, BRANCH / VALU = %.3f" % (med["SQ_INSTS_SALU"] / med["SQ_INSTS_VALU"],
                                                                    med.get("SQ_INSTS_BRANCH", 0) / med["SQ_INSTS_VALU"]))
PY
