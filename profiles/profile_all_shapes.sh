#!/usr/bin/env bash
# rocprofv3 kernel stats for the four BASELINE per-GPU shapes, one process per shape and launch mode (under rocprofv3 the
# memory an arena hands back is not returned to the driver - profiles/r03_place/arena_memory_check.py under the profiler:
# 137 GB of 2 x 64 GiB never come back - so a profiled process can place ONE buffer well; un-profiled processes place any number).
# Usage (on the GPU box, repo root): bash profiles/profile_all_shapes.sh <tag>
TAG="${1:-r01g}"
R=$PWD
export TMPDIR=/tmp
rm -rf "$R/gpurun_out/shapes_${TAG}"
# (9 x 4096 moves 10 MB per step: 128 steps per launch make its rollout buffer 1 GB, i.e. an HBM stream; the other shapes 20 / 4)
for cfg in "27 4096 400 40 20" "9 4096 1024 128 128" "81 2048 200 20 20" "243 8192 40 8 20"; do
  set -- $cfg
  for mode in rollout step; do
    O=$R/gpurun_out/shapes_${TAG}/n$1_$mode
    mkdir -p $O
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --agents $1 --envs $2 --steps $3 --warmup $4 --chunk $5 --mode $mode --no-cpu-baseline --no-extra > $O/bench.json 2> $O/bench.err)
    echo "profiled N=$1 B=$2 $mode"
  done
done
python3 $R/profiles/summarize_shapes.py "$R/gpurun_out/shapes_${TAG}" "$R/gpurun_out/${TAG}_all_shapes.md"
