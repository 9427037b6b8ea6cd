#!/usr/bin/env bash
# 9 agents x 4096 envs with an HBM-size rollout buffer (128 steps per launch, 1.02 GB): which geometry of the pipelined
# kernel?  Experiment builds build/exp/libfg_r16_<v>.so force one (1: 8 envs/wg + tile writer, 2: 16 envs + tiles,
# 3: 4 envs + tiles, 4: 8 envs + rows writer with 4 writer waves); base = 4 envs per workgroup, rows writer.
set -u
cd "$(dirname "$0")/.."
LOG=gpurun_out/r04_r9_hbm.txt; : > $LOG
for rep in 1 2; do
  for lib in base 1 2 3 4; do
    if [ $lib = base ]; then L=""; else L=build/exp/libfg_r16_$lib.so; fi
    echo -n "arm $lib: " >> $LOG
    FG_AB_DIGEST=1 FG_EXPERIMENT_LIB=$L timeout -k 10 120 python3 profiles/r03_rollout_ab.py 9:4096:128 16:4096:60 2>&1 | grep -v amdgpu.ids | tr '\n' '|' >> $LOG; echo >> $LOG
  done
done
for g in 70 140 206; do
  echo -n "243 x 8192 x 4, arena <= $g GB: " >> $LOG
  FG_AB_ARENA_GB=$g timeout -k 10 200 python3 profiles/r03_rollout_ab.py 243:8192:4 2>&1 | grep -v amdgpu.ids >> $LOG
done
cat $LOG
