#!/usr/bin/env bash
# A/B helper: ab.sh "<agents> <envs> <T,E> <steps> <chunk> <writer>" ...   (prints one line per config)
for cfg in "$@"; do
  set -- $cfg
  FG_FLAT=$6 FG_GEOM=$3 python bench.py --agents $1 --envs $2 --steps $4 --warmup 20 --chunk $5 --no-cpu-baseline 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('wr=$6', d['config']['kernel'], 'B=$2 step us', d['roofline']['avg_launch_us'], 'GB/s', d['roofline']['achieved'], '| rollout GB/s', d['other_mode']['achieved_GBps'])"
done
