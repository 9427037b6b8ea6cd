#!/usr/bin/env bash
# Round 3: counter evidence for the rollout kernels of ALL per-GPU BASELINE shapes: kernel trace,
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes), partial-write ratio (TCC_EA0_WRREQ vs _64B), store-side
# stalls (SQ / TCC), each pass around the same short bench command.  On the GPU box, from the repo root:
#     bash profiles/r03_wide_pmc.sh <tag> ["N B chunk steps" ...]
# then  python3 profiles/summarize.py gpurun_out/prof_<tag>_<N>x<B> profiles/<tag>_<N>x<B>_rollout
set -u
TAG="${1:-r03}"; shift || true
if [ $# -eq 0 ]; then set -- "81 2048 20 200" "243 8192 4 24" "27 4096 20 300"; fi
R=$PWD
export TMPDIR=/tmp
PASSES=("FETCH_SIZE" "WRITE_SIZE"
        "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_WRITE_sum"
        "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
        "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
        "TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
        "TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_REQ_sum TCC_NORMAL_WRITEBACK_sum")
NAMES=(fetch write wrreq sq sqvmem tccbusy tccstall)
for cfg in "$@"; do
  set -- $cfg
  N=$1; B=$2; CH=$3; ST=$4
  OUT="$R/gpurun_out/prof_${TAG}_${N}x${B}"
  rm -rf "$OUT"; mkdir -p "$OUT"
  BENCH="python3 $R/bench.py --agents $N --envs $B --chunk $CH --steps $ST --warmup $((ST / 5)) --no-cpu-baseline --no-extra"
  cd /tmp
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace pass failed for $cfg"; cd "$R"; exit 1; }
  i=0
  for pass in "${PASSES[@]}"; do
    timeout -k 10 240 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_${NAMES[$i]}" -- $BENCH > "$OUT/pmc_${NAMES[$i]}.log" 2>&1 || { echo "pmc pass ${NAMES[$i]} failed for $cfg"; cd "$R"; exit 1; }
    i=$((i + 1))
  done
  cd "$R"
  # un-profiled bench line of the same build on the same box (never compare profiled with un-profiled timings)
  timeout -k 10 240 python3 bench.py --agents $N --envs $B --chunk $CH --steps $ST --warmup $((ST / 5)) --no-cpu-baseline --no-extra > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed for $cfg"; exit 1; }
  python3 profiles/summarize.py "$OUT" "$R/gpurun_out/${TAG}_${N}x${B}_rollout" > /dev/null
  # keep what is committed small: the stats CSV + the summary; raw traces stay in gpurun_out
  cp "$OUT"/trace/*/*_kernel_stats.csv "$R/gpurun_out/${TAG}_${N}x${B}_rollout_kernel_stats.csv" 2>/dev/null
  find "$OUT" -name '*_kernel_trace.csv' -delete; find "$OUT" -name '*_counter_collection.csv' -size +8M -delete
  echo "profiled $cfg"
done
