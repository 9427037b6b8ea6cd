#!/usr/bin/env bash
# A/B helper: ab.sh "<agents> <envs> <T,E> <steps> <chunk> <writer>" ...   (prints one line per config)
for cfg in "$@"; do
  set -- $cfg
  FG_FLAT=$6 FG_GEOM=$3 python bench.py --agents $1 --envs $2 --steps $4 --warmup 20 --chunk $5 --no-cpu-baseline 2>/dev/null | \
    python -c "
import sys, json
d = json.loads(sys.stdin.readline())
m = {d['config']['mode']: (d['ms_per_step'] * 1e3, d['roofline']['achieved']),
     d['other_mode']['mode']: (d['other_mode']['ms_per_step'] * 1e3, d['other_mode']['achieved_GBps'])}
print('wr=$6 N=$1 B=$2 geom=$3 | step %.2f us/step %.0f GB/s | rollout %.2f us/step %.0f GB/s' % (m['step'] + m['rollout']))"
done
