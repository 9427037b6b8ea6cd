#!/usr/bin/env bash
# Counter passes around the same 27 x 4096 x 20 rollout launch into fast and slow compositions of one arena.
#   bash profiles/r05_place_channels.sh [passes...]   (GPU box, repo root) -> gpurun_out/r05_place/
# Every pass is a fresh process (the TCC block has 4 counter slots; physical memory differs between processes, so each pass
# carries its own durations: kernel trace + the launch-side HIP events in the log).
set -u
R=$PWD
export TMPDIR=/tmp
OUT=$R/gpurun_out/r05_place; mkdir -p $OUT
declare -A PASS
PASS[none]=""
PASS[wr]="TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_BUSY"
PASS[tag]="TCC_TAG_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ_LEVEL TCC_CYCLE"
PASS[wr64]="TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_DRAM TCC_EA0_WRREQ_GMI_CREDIT_STALL TCC_EA0_WRREQ_IO_CREDIT_STALL"
PASS[tcp]="TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
PASS[utcl]="TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
PASS[lat]="TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_WRITE_REQ TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS"
PASS[ta]="TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_TA_BUSY TCP_WRITE_TAGCONFLICT_STALL_CYCLES"
PASS[rd]="TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM TCC_REQ TCC_WRITEBACK"
for p in "${@:-none wr}"; do
  for q in $p; do
    rm -rf $OUT/$q; mkdir -p $OUT/$q
    if [ -z "${PASS[$q]}" ]; then
      (cd /tmp && timeout -k 10 170 python3 $R/profiles/r05_place_channels.py $OUT/$q/cands.json ${FG_PC_SHAPE:-27 4096 20} ${FG_PC_ARENA_GB:-} > $OUT/$q/run.log 2>&1) || echo "pass $q failed"
    else
      (cd /tmp && timeout -k 10 280 rocprofv3 --kernel-trace --pmc ${PASS[$q]} --output-format csv json -d $OUT/$q/prof -- python3 $R/profiles/r05_place_channels.py $OUT/$q/cands.json ${FG_PC_SHAPE:-27 4096 20} ${FG_PC_ARENA_GB:-} > $OUT/$q/run.log 2>&1) || { echo "pass $q failed"; tail -5 $OUT/$q/run.log; }
    fi
    grep -v amdgpu.ids $OUT/$q/run.log | tail -30
    # keep what the join needs, drop the bulky rest (gpurun merges <= 64 MiB back)
    find $OUT/$q/prof -name '*.json' -size +40M -delete 2>/dev/null
    ls -la $(find $OUT/$q -type f) 2>/dev/null | awk '{print $5, $9}'
  done
done
