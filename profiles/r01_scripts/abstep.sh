#!/usr/bin/env bash
# A/B of per-step kernel shapes: abstep.sh <agents> <envs> "<T,E> <writer>" ...
N=$1; B=$2; shift 2
for cfg in "$@"; do
  set -- $cfg
  FG_GEOM=$1 FG_FLAT=$2 python bench.py --mode step --agents $N --envs $B --steps 1000 --warmup 100 --no-cpu-baseline --no-extra 2>/dev/null | \
    python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('N=$N B=$B geom=$1 wr=$2 | step %.3f us/step %.0f GB/s  [%s]' % (d['ms_per_step'] * 1e3, d['roofline']['achieved'], d['config']['kernel']))"
done
