#!/usr/bin/env bash
# Both bench modes on the four formation_hd_env shapes of BASELINE.json (per GPU), one line each.
for cfg in "27 4096 1000" "9 4096 2000" "81 2048 400" "243 8192 100"; do
  set -- $cfg
  python bench.py --agents $1 --envs $2 --steps $3 --warmup $(( $3 / 10 )) --no-cpu-baseline 2>/dev/null | \
    python -c "
import sys, json
d = json.loads(sys.stdin.readline())
o = d['other_mode']
print('N=$1 B=$2 | rollout %.2f us/step %.0f GB/s (%.1f %%) %.3g env-steps/s | step %.2f us/step %.0f GB/s' % (d['ms_per_step'] * 1e3, d['roofline']['achieved'], 100 * d['roofline']['frac'], d['value'], o['ms_per_step'] * 1e3, o['achieved_GBps']))"
done
