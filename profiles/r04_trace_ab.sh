#!/usr/bin/env bash
# Cycle stamps inside the pipelined rollout kernel (build/exp/libfg_pace.so, FG_EXP_TRACE): per step, how long do the
# producer wave and the first / last writer wave of a workgroup work before they reach the step's barrier - with the
# rollout buffer in the Infinity Cache (20 steps) and in HBM (128 steps), same kernel instantiation (FG_EXP_FORCE16)?
set -u
cd "$(dirname "$0")/.."
LOG=gpurun_out/r04_trace_ab.txt; : > $LOG
for shape in 9:4096:20 9:4096:128 16:4096:20 16:4096:60 27:4096:20 8:8192:20 8:8192:60; do
  echo "== $shape" >> $LOG
  FG_EXP_TRACE=1 FG_EXP_FORCE16=1 FG_EXPERIMENT_LIB=build/exp/libfg_pace.so timeout -k 10 300 python3 profiles/r03_rollout_ab.py $shape 2>&1 | grep -v amdgpu.ids | sed 's/probe \[[^]]*\]//' >> $LOG || exit 1
done
cat $LOG
