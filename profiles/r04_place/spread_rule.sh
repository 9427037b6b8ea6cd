#!/usr/bin/env bash
# one fresh process per configuration (the physical layout of a process's memory is what is being measured)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_place; mkdir -p $OUT
LOG=$OUT/spread_rule.txt; : > $LOG
for rep in 1 2; do
for cfg in "27 4096 20" "81 2048 20"; do
  for rs in "2 0" "2 4" "2 16" "2 48" "3 0" "3 8" "3 24" "3 64" "4 16"; do
    timeout -k 10 120 python3 profiles/r04_place/spread_rule.py $cfg $rs >> $LOG 2>&1 || echo "FAILED $cfg $rs" >> $LOG
  done
done
done
timeout -k 10 120 python3 profiles/r04_place/spread_rule.py 27 4096 20 2 48 32 >> $LOG 2>&1
timeout -k 10 120 python3 profiles/r04_place/spread_rule.py 27 4096 20 3 24 32 >> $LOG 2>&1
timeout -k 10 120 python3 profiles/r04_place/spread_rule.py 27 4096 20 2 48 64 >> $LOG 2>&1
timeout -k 10 120 python3 profiles/r04_place/spread_rule.py 243 8192 4 2 48 >> $LOG 2>&1
timeout -k 10 120 python3 profiles/r04_place/spread_rule.py 243 8192 4 3 0 >> $LOG 2>&1
cat $LOG
