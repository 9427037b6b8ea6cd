#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_place; mkdir -p $OUT
LOG=$OUT/prior_alloc.txt; : > $LOG
for rep in 1 2; do
  for pre in 0 1.5 8 40 120; do
    for arena in 0 32 64 192; do
      timeout -k 10 120 python3 profiles/r04_place/prior_alloc.py $pre $arena 2>&1 | grep -v amdgpu.ids >> $LOG
    done
  done
done
for pre in 0 1.5 40; do for arena in 0 64; do
  timeout -k 10 120 python3 profiles/r04_place/prior_alloc.py $pre $arena 81 2048 20 2>&1 | grep -v amdgpu.ids >> $LOG
done; done
cat $LOG
