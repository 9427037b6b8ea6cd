#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_place; mkdir -p $OUT
LOG=$OUT/diag_repeat.txt; : > $LOG
for rep in 1 2 3 4; do
  timeout -k 10 120 python3 profiles/r04_place/default_api_diag.py 2>&1 | grep -v amdgpu.ids | cut -c1-420 >> $LOG
  timeout -k 10 120 python3 profiles/r04_place/prior_alloc.py 0 0 2>&1 | grep -v amdgpu.ids >> $LOG
done
cat $LOG
