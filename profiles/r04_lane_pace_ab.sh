#!/usr/bin/env bash
# The lane kernels' writer wave with idle cycles after every 1 KiB store (64 x FG_EXP_LANE_PACE cycles), interleaved with
# the shipped form: is the store stream of the landmark scenarios held back by bursts, as the 27-agent one was?
set -u
cd "$(dirname "$0")/.."
LOG=gpurun_out/r04_lane_pace_ab.txt; : > $LOG
for rep in 1 2; do
  for lib in base lanepace1 lanepace2 lanepace4; do
    if [ $lib = base ]; then L=""; else L=build/exp/libfg_$lib.so; fi
    echo "== arm $lib" >> $LOG
    FG_EXPERIMENT_LIB=$L timeout -k 10 300 python3 profiles/r04_scenario_rollout.py 2>&1 | grep "^| [bf]" | cut -d'|' -f2,8,9,10,12 >> $LOG || exit 1
    FG_EXPERIMENT_LIB=$L timeout -k 10 300 python3 profiles/r03_rollout_ab.py 3:65536:60 4:65536:40 2>&1 | grep -v amdgpu.ids | sed 's/probe \[[^]]*\]//' >> $LOG || exit 1
  done
done
cat $LOG
