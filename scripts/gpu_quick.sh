#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=line 2>&1 | tail -5
timeout 600 python scripts/perf_probe.py > gpurun_out/perf_probe.log 2>&1
grep -E "conv_nn|gemm_nt|clip" gpurun_out/perf_probe.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-400
