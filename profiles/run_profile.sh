#!/usr/bin/env bash
# Profiling recipe used for the summaries committed in this directory (run on the GPU box
# from the repo root through gpurun).  $1 = output tag, e.g. r01
set -u
TAG="${1:-r01}"
OUT="$PWD/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 300 --warmup 60 --no-cpu-baseline --no-extra"
cd /tmp
# 1. per-kernel time
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
# 2. HBM traffic counters, each in its own pass (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1
cd - > /dev/null
# 3. un-profiled bench line of the same build (never compare profiled with un-profiled timings)
python3 bench.py --steps 1000 --warmup 100 > "$OUT/bench.json" 2> "$OUT/bench.err"
find "$OUT" -name '*.csv' | head -20
