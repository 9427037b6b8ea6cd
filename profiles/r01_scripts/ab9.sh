#!/usr/bin/env bash
# A/B of the 9-agent (and 3-agent) rollout shapes: ab9.sh <agents> <envs> <chunk> <variant...>
N=$1; B=$2; C=$3; shift 3
for v in "$@"; do
  FG_ROLL9=$v python bench.py --agents $N --envs $B --steps 2000 --warmup 200 --chunk $C --no-cpu-baseline --no-extra 2>/dev/null | \
    python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('N=$N B=$B chunk=$C v=$v | rollout %.3f us/step %.0f GB/s' % (d['ms_per_step'] * 1e3, d['roofline']['achieved']))"
done
