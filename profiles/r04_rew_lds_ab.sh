#!/usr/bin/env bash
# Producers of the pipelined rollout kernel without global stores (rewards / done flags go through LDS to the writer
# waves) and without a possible load from the device RNG counter inside the step loop - interleaved against the previous
# build (build/exp/libfg_head.so) and against the counter hoist alone (build/exp/libfg_hoist.so).
set -u
cd "$(dirname "$0")/.."
LOG=gpurun_out/r04_rew_lds_ab.txt; : > $LOG
for rep in 1 2; do
  for lib in head hoist new; do
    if [ $lib = new ]; then L=""; else L=build/exp/libfg_$lib.so; fi
    echo "== arm $lib" >> $LOG
    FG_AB_DIGEST=1 FG_EXPERIMENT_LIB=$L timeout -k 10 300 python3 profiles/r03_rollout_ab.py 9:4096:128 9:4096:20 16:4096:60 27:4096:20 8:8192:60 25:4096:20 32:4096:20 64:2048:20 3:1024:20 9:1024:20 2>&1 | grep -v amdgpu.ids | sed 's/probe \[[^]]*\]//' >> $LOG || exit 1
  done
done
for lib in head new head new; do
  if [ $lib = new ]; then L=""; else L=build/exp/libfg_$lib.so; fi
  echo "== scenarios, arm $lib" >> $LOG
  FG_EXPERIMENT_LIB=$L timeout -k 10 300 python3 profiles/r04_scenario_rollout.py 2>&1 | grep "^| [bf]" | cut -d'|' -f2,8,9,10,12 >> $LOG || exit 1
done
cat $LOG
