#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -120 > gpurun_out/gpu_tests.log
grep -E "^E  |passed|failed|^FAILED" gpurun_out/gpu_tests.log | head -40
timeout 600 python scripts/perf_probe.py > gpurun_out/perf_probe.log 2>&1
grep -E "conv_nn|gemm_nt|clip" gpurun_out/perf_probe.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log
