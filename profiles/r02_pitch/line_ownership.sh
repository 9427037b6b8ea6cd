#!/usr/bin/env bash
# A/B of the line-ownership tile writer (new library = the in-tree build, old = $1, a libformation_hip.so built from
# the commit before): interleaved rounds inside one gpurun call.
set -u
OLD=$1
OUT=gpurun_out/r02_pitch; mkdir -p $OUT
cp gym-formation_amd/lib/libformation_hip.so $OUT/lib_new.so
line() { python3 -c "
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][0]); t=d['timing']
print('%-46s us/step %.3f  GB/s %.0f  frac %.4f  blocks %d min/med/max %.3f/%.3f/%.3f' % ('$2', d['ms_per_step']*1e3, d['roofline']['achieved'], d['roofline']['frac'], t['blocks'], t['block_ms_min'], t['block_ms_median'], t['block_ms_max']))"; }
for r in 1 2 3; do
  for shape in "27 4096" "27 16384" "27 65536"; do
    for v in old new; do
      set -- $shape $v
      if [ $3 = old ]; then cp $OLD gym-formation_amd/lib/libformation_hip.so; else cp $OUT/lib_new.so gym-formation_amd/lib/libformation_hip.so; fi
      python3 bench.py --agents $1 --envs $2 --steps 400 --warmup 40 --no-extra --no-cpu-baseline > $OUT/b.json 2>/dev/null
      line $OUT/b.json "round $r  $1 x $2  lib $3" | tee -a $OUT/line_ownership4.txt
    done
  done
done
cp $OUT/lib_new.so gym-formation_amd/lib/libformation_hip.so
