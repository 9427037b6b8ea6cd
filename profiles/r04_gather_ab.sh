#!/usr/bin/env bash
# The gather writer (fg_obs_writers.hpp: no LDS image, operands from a per-workgroup table) against the tile writer of the
# previous build (build/exp/libfg_base.so) at 9 and 8 agents, interleaved, digests of every output compared.
set -u
cd "$(dirname "$0")/.."
LOG=gpurun_out/r04_gather_ab.txt; : > $LOG
for rep in 1 2; do
  for lib in base new; do
    if [ $lib = new ]; then L=""; else L=build/exp/libfg_$lib.so; fi
    echo "== arm $lib" >> $LOG
    FG_AB_DIGEST=1 FG_EXPERIMENT_LIB=$L timeout -k 10 300 python3 profiles/r03_rollout_ab.py 9:4096:128 9:6000:100 9:8192:64 9:16384:32 9:65536:8 8:4096:120 8:8192:60 8:65536:20 2>&1 | grep -v amdgpu.ids | sed 's/probe \[[^]]*\]//' >> $LOG || exit 1
  done
done
cat $LOG
