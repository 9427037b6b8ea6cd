#!/usr/bin/env bash
# action prefetch four steps ahead (build/exp/libfg_pf4.so) vs one step ahead (the shipped library of the moment), interleaved
set -u
cd "$(dirname "$0")/.."
LOG=gpurun_out/r04_pf4_ab.txt; : > $LOG
for rep in 1 2 3; do
  for lib in base pf4; do
    if [ $lib = base ]; then L=""; else L=build/exp/libfg_pf4.so; fi
    echo -n "arm $lib: " >> $LOG
    FG_AB_DIGEST=1 FG_EXPERIMENT_LIB=$L timeout -k 10 200 python3 profiles/r03_rollout_ab.py 9:4096:128 9:4096:20 3:65536:60 27:4096:20 16:8192:20 2>&1 | grep -v amdgpu.ids | sed 's/ steps per launch//; s/probe \[[^]]*\]//' | tr '\n' '|' >> $LOG; echo >> $LOG
  done
done
cat $LOG
