#!/usr/bin/env bash
# Is the writer of an HBM-size rollout held up by address translation (every step writes a new slab = new pages for every
# workgroup)?  build/exp/libfg_pace.so, FG_EXP_WARM = w: the producer lane of every w-th env reads one float of the
# observation slab of step k + 2 (consumed a step later), so that the translation is cached when the writers get there.
set -u
cd "$(dirname "$0")/.."
LOG=gpurun_out/r04_warm_ab.txt; : > $LOG
for rep in 1 2; do
  for warm in 0 1 2 4 16; do
    echo "== warm $warm" >> $LOG
    FG_EXP_WARM=$warm FG_AB_DIGEST=1 FG_EXPERIMENT_LIB=build/exp/libfg_pace.so timeout -k 10 300 python3 profiles/r03_rollout_ab.py 9:4096:128 8:8192:60 16:4096:60 27:4096:20 25:4096:20 2>&1 | grep -v amdgpu.ids | sed 's/probe \[[^]]*\]//' >> $LOG || exit 1
  done
done
cat $LOG
