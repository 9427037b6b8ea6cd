#!/usr/bin/env bash
# Actions loaded four steps at a time, once per four steps (ACT4 in rollout_kernel: workgroups of more than 512 threads and
# the gather-writer instantiations) against one step ahead every step (the previous build, build/exp/libfg_base.so):
# fewer, larger read events between the store streams.  Interleaved, digests of every output compared.
set -u
cd "$(dirname "$0")/.."
LOG=gpurun_out/r04_batch4_ab.txt; : > $LOG
for rep in 1 2; do
  for lib in base new; do
    if [ $lib = new ]; then L=""; else L=build/exp/libfg_$lib.so; fi
    echo "== arm $lib" >> $LOG
    FG_AB_DIGEST=1 FG_EXPERIMENT_LIB=$L timeout -k 10 400 python3 profiles/r03_rollout_ab.py 27:4096:20 27:16384:5 27:2048:20 27:1024:20 9:4096:128 9:8192:64 9:16384:32 9:65536:8 9:4096:20 9:2048:250 8:8192:60 8:65536:20 8:4096:120 16:4096:60 16:8192:30 25:4096:20 32:4096:20 64:2048:20 3:1024:20 2>&1 | grep -v amdgpu.ids | sed 's/probe \[[^]]*\]//' >> $LOG || exit 1
  done
done
cat $LOG
