#!/usr/bin/env bash
# the lane kernels' writer wave with 16-byte stores (build/exp/libfg_lane_x4.so) vs 8-byte stores, interleaved
set -u
cd "$(dirname "$0")/.."
LOG=gpurun_out/r04_lane_x4_ab.txt; : > $LOG
for rep in 1 2; do
  for lib in base x4; do
    if [ $lib = base ]; then L=""; else L=build/exp/libfg_lane_x4.so; fi
    echo "== arm $lib" >> $LOG
    FG_EXPERIMENT_LIB=$L timeout -k 10 300 python3 profiles/r04_scenario_rollout.py 2>&1 | grep "^| [bf]" | cut -d'|' -f2,8,9,10,12 >> $LOG
    FG_AB_DIGEST=1 FG_EXPERIMENT_LIB=$L timeout -k 10 300 python3 profiles/r03_rollout_ab.py 3:65536:60 4:65536:40 2>&1 | grep -v amdgpu.ids | sed 's/probe \[[^]]*\]//' >> $LOG
  done
done
cat $LOG
