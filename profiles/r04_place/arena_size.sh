#!/usr/bin/env bash
# How large does the probe's arena have to be?  The r03 probe (>= 8 spread selections timed with the env's own launch) with its
# arena capped at G GB; interleaved rounds, one fresh process per arm.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_place; mkdir -p $OUT
LOG=$OUT/arena_size.txt; : > $LOG
for rep in 1 2 3; do
  for shape in 27:4096:20 81:2048:20; do
    echo -n "no probe: " >> $LOG
    FG_AB_CANDIDATES=1 timeout -k 10 120 python3 profiles/r03_rollout_ab.py $shape 2>&1 | grep -v amdgpu.ids >> $LOG
    for g in 4 8 16 32 64 192; do
      echo -n "arena <= $g GB: " >> $LOG
      FG_AB_ARENA_GB=$g timeout -k 10 120 python3 profiles/r03_rollout_ab.py $shape 2>&1 | grep -v amdgpu.ids >> $LOG
    done
  done
done
cat $LOG
