#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -120 > gpurun_out/gpu_tests.log
grep -E "^E  |passed|failed|^FAILED" gpurun_out/gpu_tests.log | head -40
bash scripts/gpu_profile.sh r1v1
head -12 gpurun_out/prof_r1v1/trace/bench_kernel_stats.csv | cut -c1-200
for f in gpurun_out/prof_r1v1/pmc_*/bench_counter_collection.csv; do echo $f; head -3 $f | cut -c1-300; done
