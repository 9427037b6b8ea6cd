#!/usr/bin/env bash
# EXPERIMENT: workgroups of the 27-agent rollout kernel started in phases (build/exp/libfg_stagger_S_TICKS.so) vs lockstep,
# on plain allocations (FG_AB_CANDIDATES=1) and on probe-placed buffers; interleaved rounds, one process per arm
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_place; mkdir -p $OUT
LOG=$OUT/stagger_ab.txt; : > $LOG
for cand in 1 8; do
for rep in 1 2 3; do
  for lib in base 2_600 4_900 4_1800 8_1200; do
    if [ $lib = base ]; then L=""; else L=build/exp/libfg_stagger_$lib.so; fi
    echo -n "candidates $cand arm $lib: " >> $LOG
    FG_AB_DIGEST=1 FG_AB_CANDIDATES=$cand FG_EXPERIMENT_LIB=$L timeout -k 10 120 python3 profiles/r03_rollout_ab.py 27:4096:20 2>&1 | grep -v amdgpu.ids >> $LOG
  done
done
done
cat $LOG
