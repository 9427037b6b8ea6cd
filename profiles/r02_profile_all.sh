#!/usr/bin/env bash
# Round-2 profile set, one gpurun call (repo root):  bash profiles/r02_profile_all.sh
#   1. headline workload: rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
#      then an un-profiled bench line                         -> gpurun_out/prof_r02 (profiles/run_profile.sh)
#   2. all BASELINE per-GPU shapes, both launch modes         -> gpurun_out/shapes_r02 (profiles/profile_all_shapes.sh)
#   3. controller / scenario / reset kernels                  -> gpurun_out/r02_aux
set -u
export TMPDIR=/tmp
R=$PWD
bash profiles/run_profile.sh r02 > gpurun_out/r02_run_profile.log 2>&1
python3 profiles/summarize.py gpurun_out/prof_r02 gpurun_out/r02_27x4096_rollout > /dev/null 2>&1
cp gpurun_out/prof_r02/trace/*/*_kernel_stats.csv gpurun_out/r02_27x4096_rollout_kernel_stats.csv 2>/dev/null
bash profiles/profile_all_shapes.sh r02 > gpurun_out/r02_shapes.log 2>&1
mkdir -p gpurun_out/r02_aux
python3 profiles/r02_aux_kernels.py > gpurun_out/r02_aux/aux_kernels.md 2> gpurun_out/r02_aux/err.log
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_aux/trace -- python3 $R/profiles/r02_aux_kernels.py > /dev/null 2>&1)
cp gpurun_out/r02_aux/trace/*/*_kernel_stats.csv gpurun_out/r02_aux/aux_kernel_stats.csv 2>/dev/null
ls gpurun_out/r02_aux gpurun_out/prof_r02 | head -30
tail -20 gpurun_out/r02_aux/aux_kernels.md
