#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -x -k "conv" 2>&1 | tail -3
echo "== wide"; timeout 600 python scripts/probe_gemm.py f32x3 2>&1 | tail -7 | head -4
[ -f brainmagick_amd/libbmhip_prof.so ] && BM_HIP_LIB=$PWD/brainmagick_amd/libbmhip_prof.so timeout 200 python scripts/prof_gemm_wide.py 2>&1 | grep -E "dil|wave [01]"
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-200
