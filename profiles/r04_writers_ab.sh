#!/usr/bin/env bash
# The trace (r04_trace_ab.txt) shows the tile writer, not the producers or the memory, bounding 9- and 8-agent rollouts into
# HBM-size buffers (writer wave 5400-5700 cycles per step against 2800 for the producers).  More writer waves per env?
# build/exp/libfg_pace.so, FG_EXP_GEOM: 9 agents 0 = <TP256,TW256,E16>, 1 = <256,512,16>, 2 = <128,256,8>, 3 = <128,512,8>,
# 4 = <64,256,4>; 8 agents 0 = <64,128,E8>, 1 = <64,256,8>, 2 = <128,256,16>, 3 = <128,512,16>, 4 = <64,512,8>
set -u
cd "$(dirname "$0")/.."
LOG=gpurun_out/r04_writers_ab.txt; : > $LOG
for rep in 1 2; do
  for g in 0 1 2 3 4; do
    echo "== geom $g" >> $LOG
    FG_EXP_GEOM=$g FG_AB_DIGEST=1 FG_EXPERIMENT_LIB=build/exp/libfg_pace.so timeout -k 10 300 python3 profiles/r03_rollout_ab.py 9:4096:128 9:8192:64 9:16384:32 9:65536:8 8:4096:120 8:8192:60 8:65536:20 2>&1 | grep -v amdgpu.ids | sed 's/probe \[[^]]*\]//' >> $LOG || exit 1
  done
done
cat $LOG
