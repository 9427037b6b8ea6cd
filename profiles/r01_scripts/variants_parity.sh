#!/usr/bin/env bash
# every tuning switch selects among variants that must stay parity-green: run the step / rollout parity tests under each
set -u
K='teacher_forced or split_stages or rollout_equals or auto_reset_equals or full_size or rollout_api or baseline_full_size'
for cfg in "FG_FLAT=1" "FG_FLAT=10" "FG_FLAT=4" "FG_GEOM=128,4" "FG_GEOM=256,8" "FG_ROLLWR=0" "FG_ROLLE=8" "FG_ROLLE=4" "FG_ROLLE=2" "FG_TW=512" "FG_TW=128" \
           "FG_NOPIPE=1" "FG_PIPE81=1" "FG_SHARE=1" "FG_ROLL9=0" "FG_ROLL9=1" "FG_ROLL9=4" "FG_ROLL9=6" "FG_STEPWG=128"; do
  res=$(env $cfg timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -1)
  echo "$cfg : $res"
done
