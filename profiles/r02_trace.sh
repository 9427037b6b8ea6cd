#!/usr/bin/env bash
set -u
OUT=gpurun_out/r02_trace; mkdir -p $OUT
cp gym-formation_amd/lib/libformation_hip.so $OUT/lib_keep.so
FG_EXTRA_FLAGS="-DFG_TRACE" bash gym-formation_amd/csrc/build.sh > $OUT/build.log 2>&1
for cfg in "81 2048" "27 4096" "9 4096" "243 1024"; do
  set -- $cfg
  python3 profiles/r02_trace.py $1 $2 2>/dev/null | tee -a $OUT/trace.txt
  echo | tee -a $OUT/trace.txt
done
cp $OUT/lib_keep.so gym-formation_amd/lib/libformation_hip.so
