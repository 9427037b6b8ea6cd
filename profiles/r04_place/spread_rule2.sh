#!/usr/bin/env bash
# spacer kinds (hipMalloc / VMM chunks created / created + mapped), one fresh process per configuration, three interleaved rounds
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_place; mkdir -p $OUT
LOG=$OUT/spread_rule2.txt; : > $LOG
for rep in 1 2 3; do
for cfg in "27 4096 20" "81 2048 20"; do
  for mode in 0 1 2; do
    for rs in "2 16" "2 48" "3 16"; do
      timeout -k 10 120 python3 profiles/r04_place/spread_rule.py $cfg $rs 0 $mode 2>&1 | grep -v amdgpu.ids >> $LOG || echo "FAILED $cfg $rs $mode" >> $LOG
    done
  done
done
done
cat $LOG
