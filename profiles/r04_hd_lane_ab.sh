#!/usr/bin/env bash
# formation_hd_env rollouts at 3 and 4 agents: one env per lane (fg_hd_lane_kernel.hpp; build/exp/libfg_hdlane_all.so takes it at
# every batch size) against a lane per agent (fg::rollout_kernel; libfg_hdlane_none.so never takes the lane kernel), interleaved;
# digests must agree
set -u
cd "$(dirname "$0")/.."
LOG=gpurun_out/r04_hd_lane_ab.txt; : > $LOG
for rep in 1 2; do
  for lib in none all; do
    echo -n "arm $lib: " >> $LOG
    FG_AB_DIGEST=1 FG_EXPERIMENT_LIB=build/exp/libfg_hdlane_$lib.so timeout -k 10 300 python3 profiles/r03_rollout_ab.py 3:1024:20 3:4096:20 3:16384:20 3:65536:60 3:262144:20 4:4096:20 4:16384:20 4:65536:40 2>&1 | grep -v amdgpu.ids | sed 's/ steps per launch//; s/probe \[[^]]*\]//; s/(min [0-9.]* max [0-9.]*)//; s/ [0-9]* GB\/s//' | tr '\n' '|' >> $LOG; echo >> $LOG
  done
done
cat $LOG
