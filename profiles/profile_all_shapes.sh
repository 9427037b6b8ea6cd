#!/usr/bin/env bash
# rocprofv3 kernel stats for the four BASELINE per-GPU shapes, both launch modes in one process each.
# Usage (on the GPU box, repo root): bash profiles/profile_all_shapes.sh <tag>
TAG="${1:-r01g}"
R=$PWD
export TMPDIR=/tmp
for cfg in "27 4096 400 40" "9 4096 800 80" "81 2048 200 20" "243 8192 40 8"; do
  set -- $cfg
  O=$R/gpurun_out/shapes_${TAG}/n$1
  mkdir -p $O
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --agents $1 --envs $2 --steps $3 --warmup $4 --no-cpu-baseline --no-other-configs --no-small-buffer > $O/bench.json 2> $O/bench.err)
  echo "profiled N=$1 B=$2"
done
python3 $R/profiles/summarize_shapes.py "$R/gpurun_out/shapes_${TAG}" "$R/gpurun_out/${TAG}_all_shapes.md"
