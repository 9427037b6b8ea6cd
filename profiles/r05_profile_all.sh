#!/usr/bin/env bash
# The round-5 profile set of the final build in one gpurun call (from the repo root on the GPU box):
#   headline workload (kernel trace + PMC + un-profiled bench), counter passes of the rollout kernels of the other per-GPU
#   shapes (9 x 4096 at 128 steps per launch = an HBM-size buffer), kernel trace of all per-GPU shapes in both launch modes,
#   the landmark scenarios (A/B of the two kernels, counter passes), agent counts other than 3^L.
set -u
R=$PWD
bash profiles/run_profile.sh r05 > gpurun_out/r05_run_profile.log 2>&1 && python3 profiles/summarize.py gpurun_out/prof_r05 gpurun_out/r05_27x4096_rollout > /dev/null \
  && cp gpurun_out/prof_r05/trace/*/*_kernel_stats.csv gpurun_out/r05_27x4096_rollout_kernel_stats.csv || { echo "run_profile failed"; tail -5 gpurun_out/r05_run_profile.log; exit 1; }
echo "headline profile done"
bash profiles/r03_wide_pmc.sh r05 "9 4096 128 1024" "81 2048 20 200" "243 8192 4 24" > gpurun_out/r05_wide_pmc.log 2>&1 || { echo "wide pmc failed"; tail -5 gpurun_out/r05_wide_pmc.log; exit 1; }
echo "wide pmc done"
bash profiles/profile_all_shapes.sh r05 > gpurun_out/r05_all_shapes.log 2>&1 || { echo "all shapes failed"; tail -5 gpurun_out/r05_all_shapes.log; exit 1; }
echo "all shapes done"
python3 profiles/r04_scenario_rollout.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_scenario_rollout.md || { echo "scenario rollout failed"; exit 1; }
bash profiles/r04_scn_pmc.sh > gpurun_out/r05_scn_pmc.log 2>&1 || { echo "scn pmc failed"; tail -5 gpurun_out/r05_scn_pmc.log; exit 1; }
echo "scenarios done"
python3 profiles/r05_generic_n.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_generic_n.md || { echo "generic n failed"; exit 1; }
echo "generic n done"
