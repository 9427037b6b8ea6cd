#!/usr/bin/env bash
# Upper bound: what would the chain-bound rollouts gain if the producer waves never touched global memory?
# build/exp/libfg_nost.so = no reward / done stores from the producers; libfg_nomem.so = also no action loads (synthetic
# actions).  Not product code: the outputs are wrong by construction.
set -u
cd "$(dirname "$0")/.."
LOG=gpurun_out/r04_nomem_ab.txt; : > $LOG
for rep in 1 2; do
  for lib in hoist nost nomem; do
    echo "== arm $lib" >> $LOG
    FG_EXPERIMENT_LIB=build/exp/libfg_$lib.so timeout -k 10 300 python3 profiles/r03_rollout_ab.py 9:4096:128 9:4096:20 16:4096:60 8:8192:60 27:4096:20 3:1024:20 2>&1 | grep -v amdgpu.ids | sed 's/probe \[[^]]*\]//' >> $LOG || exit 1
  done
done
cat $LOG
