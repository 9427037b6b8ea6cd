#!/usr/bin/env bash
# 9 agents with the gather writer: workgroup geometry per batch-size class (build/exp/libfg_geom.so, FG_EXP_GEOM:
# 0 = the library's rule, 1 = <TP256,TW512,E16>, 2 = <256,256,16>, 3 = <128,256,8>, 4 = <128,128,8>, 5 = <64,128,4>,
# 6 = <64,256,4>, 7 = <64,64,4>), buffers in HBM (long launches) and in the Infinity Cache (20 steps)
set -u
cd "$(dirname "$0")/.."
LOG=gpurun_out/r04_gather_geom.txt; : > $LOG
for g in 0 1 2 3 4 5 6 7; do
  echo "== geom $g" >> $LOG
  FG_EXP_GEOM=$g FG_AB_DIGEST=1 FG_EXPERIMENT_LIB=build/exp/libfg_geom.so timeout -k 10 300 python3 profiles/r03_rollout_ab.py 9:1024:20 9:2048:250 9:4096:20 9:4096:128 9:6000:100 9:8192:20 9:8192:64 9:16384:32 9:65536:8 2>&1 | grep -v amdgpu.ids | sed 's/probe \[[^]]*\]//' >> $LOG || exit 1
done
cat $LOG
