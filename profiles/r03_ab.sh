#!/usr/bin/env bash
# Interleaved A/B rounds over experiment builds of the library (gym-formation_amd/lib/exp_<arm>.so).
#   bash profiles/r03_ab.sh "<arm> <arm> ..." "<N:B:K> ..." [rounds]
ARMS="$1"; SHAPES="$2"; ROUNDS="${3:-2}"
for r in $(seq 1 $ROUNDS); do
  for arm in $ARMS; do
    echo "== round $r arm $arm"
    FG_AB_DIGEST=1 FG_EXPERIMENT_LIB=gym-formation_amd/lib/exp_$arm.so timeout -k 10 300 python3 profiles/r03_rollout_ab.py $SHAPES || exit 1
  done
done
