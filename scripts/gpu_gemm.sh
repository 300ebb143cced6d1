#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -x -k "conv" 2>&1 | tail -12
echo "== wide"; timeout 600 python scripts/probe_gemm.py f32x3 2>&1 | tail -7
echo "== narrow"; BM_X3_WIDE=0 timeout 600 python scripts/probe_gemm.py f32x3 2>&1 | tail -7 | head -4
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-300
