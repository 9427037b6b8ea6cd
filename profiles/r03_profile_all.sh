#!/usr/bin/env bash
# The round-3 profile set of the final build in one gpurun call (from the repo root on the GPU box):
#   headline workload (kernel trace + PMC + un-profiled bench), counter passes of the wide rollout kernels, kernel trace
#   of all per-GPU shapes in both launch modes, vec-env modes / captured loops.
set -u
R=$PWD
bash profiles/run_profile.sh r03 > gpurun_out/r03_run_profile.log 2>&1 && python3 profiles/summarize.py gpurun_out/prof_r03 gpurun_out/r03_27x4096_rollout > /dev/null \
  && cp gpurun_out/prof_r03/trace/*/*_kernel_stats.csv gpurun_out/r03_27x4096_rollout_kernel_stats.csv || { echo "run_profile failed"; tail -5 gpurun_out/r03_run_profile.log; exit 1; }
echo "headline profile done"
bash profiles/r03_wide_pmc.sh r03 "81 2048 20 200" "243 8192 4 24" > gpurun_out/r03_wide_pmc.log 2>&1 || { echo "wide pmc failed"; tail -5 gpurun_out/r03_wide_pmc.log; exit 1; }
echo "wide pmc done"
bash profiles/profile_all_shapes.sh r03 > gpurun_out/r03_all_shapes.log 2>&1 || { echo "all shapes failed"; tail -5 gpurun_out/r03_all_shapes.log; exit 1; }
echo "all shapes done"
python3 profiles/r03_vec_env.py > gpurun_out/r03_vec_env.txt 2>&1 || { echo "vec env failed"; exit 1; }
echo "vec env done"
