#!/usr/bin/env bash
# steps per launch (= size of the rollout buffer that is overwritten launch after launch) x old / new tile writer
set -u
OUT=gpurun_out/r02_pitch; mkdir -p $OUT
cp gym-formation_amd/lib/libformation_hip.so $OUT/lib_new.so
for r in 1 2; do for c in 2 4 6 8 12 20 40; do for v in old new; do
  if [ $v = old ]; then cp build/lib_old.so gym-formation_amd/lib/libformation_hip.so; else cp $OUT/lib_new.so gym-formation_amd/lib/libformation_hip.so; fi
  python3 bench.py --steps 400 --warmup 40 --chunk $c --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('round $r lib $v chunk %2d (%4.0f MB)  us/step %.3f  GB/s %.0f' % ($c, $c*4096*27*162*4/1e6, d['ms_per_step']*1e3, d['roofline']['achieved']))"
done; done; done
cp $OUT/lib_new.so gym-formation_amd/lib/libformation_hip.so
