#!/usr/bin/env bash
# diagnostic: is the round-robin slowdown the observation stream or the scattered small outputs (reward / done)?
# FG_RR bit 0 = round-robin env ownership, bit 1 = skip the reward / indiv / done stores (timing only, invalid results)
set -u
OUT=gpurun_out/r02_pitch; mkdir -p $OUT
line() { python3 -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][0]); t=d['timing']
print('%-44s us/step %.3f  GB/s %.0f  blocks %d min/med/max %.3f/%.3f/%.3f' % ('$2', d['ms_per_step']*1e3, d['roofline']['achieved'], t['blocks'], t['block_ms_min'], t['block_ms_median'], t['block_ms_max']))"; }
for r in 1 2; do
  for shape in "27 4096" "27 16384"; do
    for v in "0 0" "0 2" "-1 1" "-1 3" "0 3"; do
      set -- $shape $v
      FG_RR=$4 python3 bench.py --agents $1 --envs $2 --steps 400 --warmup 40 --no-extra --no-cpu-baseline --obs-pitch $3 > $OUT/b.json 2>/dev/null
      line $OUT/b.json "round $r  $1 x $2  pitch $3  FG_RR $4" | tee -a $OUT/rr_diag.txt
    done
  done
done
