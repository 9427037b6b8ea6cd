#!/usr/bin/env bash
# r02 single-step study at 81 x 2048 (and 27 x 4096): builds library variants with FG_EXTRA_FLAGS and times
# `bench.py --mode step` for each, interleaved rounds inside one gpurun call.
set -u
OUT=gpurun_out/r02_step81; mkdir -p $OUT
variants=("base:" "prio:-DFG_EXP_PRIO")
for v in "${variants[@]}"; do
  name=${v%%:*}; flags=${v#*:}
  FG_EXTRA_FLAGS="$flags" bash gym-formation_amd/csrc/build.sh > $OUT/build_$name.log 2>&1
  cp gym-formation_amd/lib/libformation_hip.so $OUT/lib_$name.so
done
line() { python3 -c "
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][0]); t=d['timing']
print('%-40s us/step %.3f  GB/s %.0f  frac %.4f  blocks %d min/med/max %.3f/%.3f/%.3f' % ('$2', d['ms_per_step']*1e3, d['roofline']['achieved'], d['roofline']['frac'], t['blocks'], t['block_ms_min'], t['block_ms_median'], t['block_ms_max']))"; }
for r in 1 2 3; do
  for v in "${variants[@]}"; do
    name=${v%%:*}
    cp $OUT/lib_$name.so gym-formation_amd/lib/libformation_hip.so
    for shape in "81 2048" "243 1024"; do
      set -- $shape
      python3 bench.py --agents $1 --envs $2 --mode step --steps 200 --warmup 20 --no-extra --no-cpu-baseline > $OUT/b.json 2>/dev/null
      line $OUT/b.json "round $r  $name  $1 x $2 step" | tee -a $OUT/study.txt
    done
  done
done
cp $OUT/lib_base.so gym-formation_amd/lib/libformation_hip.so
