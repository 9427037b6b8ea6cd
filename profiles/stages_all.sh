#!/usr/bin/env bash
# GPU-side stage durations (physics only / step without observations / full step) for the BASELINE per-GPU shapes
export TMPDIR=/tmp
R=$PWD
for cfg in "27 4096" "9 4096" "81 2048" "243 8192"; do
  set -- $cfg
  O=$R/gpurun_out/stages_$1
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/profiles/stages.py $1 $2 > $O.log 2>&1)
  echo "== N=$1 B=$2"; python3 $R/profiles/stages_summarize.py $O
done
