#!/usr/bin/env bash
# steps per rollout launch (bench --chunk) on the headline workload, two rounds inside one gpurun call
for r in 1 2; do
for c in 10 20 30 40 60; do
python bench.py --steps 1200 --warmup 120 --chunk $c --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('chunk $c round $r | %.3f us/step %.0f GB/s' % (d['ms_per_step'] * 1e3, d['roofline']['achieved']))"
done
done
