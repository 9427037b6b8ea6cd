#!/usr/bin/env bash
# The tiles of a workgroup's 16 envs dealt round-robin over its writer waves (ONE stream of 8 consecutive 5.8 KB tiles per workgroup)
# against waves owning whole envs (8 streams 17.5 KB apart): does a more compact store window make an ORDINARY allocation run like a
# placed one?   bash profiles/r05_deal_ab.sh   (GPU box, repo root) -> gpurun_out/r05_deal_ab.txt
set -u
OUT=gpurun_out/r05_deal_ab.txt; mkdir -p gpurun_out; : > $OUT
for rnd in 1 2 3; do
  for arm in base deal; do   # (the deal arm needs a -DFG_TILE_DEAL=1 build of commit 7112a0c + the experiment patch; removed since)
    lib=gym-formation_amd/lib/libformation_hip.so; [ $arm = deal ] && lib=build/libfg_deal.so
    for cand in 1 8; do
      echo -n "round $rnd arm $arm candidates $cand: " >> $OUT
      FG_EXPERIMENT_LIB=$lib FG_AB_DIGEST=1 FG_AB_CANDIDATES=$cand FG_AB_ARENA_GB=9 timeout -k 10 120 python3 profiles/r03_rollout_ab.py 27:4096:20 2>&1 | grep -v amdgpu.ids >> $OUT
    done
  done
done
cat $OUT
