#!/usr/bin/env bash
# A/B of one tuning switch on the default bench workload inside ONE gpurun call (boxes differ by 5-10 %):
#   ab_env.sh VAR "v0 v1 ..." [rounds] [extra bench args]
VAR=$1; VALS=$2; ROUNDS=${3:-3}; shift 3 || shift $#
for r in $(seq $ROUNDS); do
  for v in $VALS; do
    env $VAR=$v python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extra "$@" 2>/dev/null | \
      python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$VAR=$v round $r | %.3f us/step %.0f GB/s  [%s]' % (d['ms_per_step'] * 1e3, d['roofline']['achieved'], d['config']['kernel']))"
  done
done
