#!/usr/bin/env bash
# Chain-bound rollouts into an HBM-size buffer: does a writer wave that spreads its stores over the step (idle cycles after
# every tile, build/exp/libfg_pace.so, FG_EXP_PACE x 128 cycles) beat one that bursts them and then waits for the producers?
set -u
cd "$(dirname "$0")/.."
LOG=gpurun_out/r04_pace_ab.txt; : > $LOG
for rep in 1 2; do
  for pace in 0 1 2 3 4 6 8; do
    echo "== pace $pace" >> $LOG
    FG_EXP_PACE=$pace FG_AB_DIGEST=1 FG_EXPERIMENT_LIB=build/exp/libfg_pace.so timeout -k 10 300 python3 profiles/r03_rollout_ab.py 9:4096:128 9:4096:20 8:8192:60 16:4096:60 2>&1 | grep -v amdgpu.ids | sed 's/probe \[[^]]*\]//' >> $LOG || exit 1
  done
done
cat $LOG
