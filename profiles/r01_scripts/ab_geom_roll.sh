#!/usr/bin/env bash
# interleaved multi-round A/B of rollout workgroup shapes at 27 x 4096 (run-to-run noise is ~3 %)
for r in 1 2 3 4; do
for cfg in "FG_ROLLE=16 FG_TW=256" "FG_ROLLE=16 FG_TW=512" "FG_ROLLE=8 FG_TW=256" "FG_ROLLE=8 FG_TW=512" "FG_ROLLE=8 FG_TW=128" "FG_ROLLE=4 FG_TW=128" "FG_ROLLE=16 FG_TW=256 FG_ROLLWR=0"; do
env $cfg python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$cfg round $r | %.3f us/step %.0f GB/s' % (d['ms_per_step'] * 1e3, d['roofline']['achieved']))"
done
done
