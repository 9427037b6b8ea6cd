#!/usr/bin/env bash
# One / two / three writer waves per 64-env workgroup of the one-env-per-lane kernels (FG_LANE_WRITERS experiment builds in build/),
# 16-byte LDS reads of contiguous block images: the landmark scenarios at 65536 envs, interleaved rounds.
#   bash profiles/r05_lane_writers_ab.sh   (GPU box, repo root) -> gpurun_out/r05_lane_writers_ab.txt
set -u
OUT=gpurun_out/r05_lane_writers_ab.txt; mkdir -p gpurun_out; : > $OUT
for rnd in 1 2; do
  for lib in build/libfg_nww1.so gym-formation_amd/lib/libformation_hip.so build/libfg_nww3.so; do
    [ -f $lib ] || continue
    echo "== round $rnd $lib" >> $OUT
    FG_EXPERIMENT_LIB=$lib timeout -k 10 200 python3 profiles/r04_scenario_rollout.py 2>&1 | grep "^| [bf]" | cut -d'|' -f2,3,4,8,9,10,12,13 >> $OUT
  done
done
cat $OUT
