#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
TAG=r4d
for U in 1 2 4 8; do
  L=$R/scripts/micro/bin/libbmhip_u$U.so; [ $U = 4 ] && L=$R/brainmagick_amd/libbmhip.so
  echo "== unroll $U"; BM_HIP_LIB=$L timeout 120 python scripts/probe_elementwise.py 2>&1 | grep -v amdgpu.ids
done
for N in 4 16 32; do echo "== unroll 4, nsplit $N"; BM_BWD_NSPLIT=$N timeout 120 python scripts/probe_elementwise.py 2>&1 | grep -E "bn_bwd|glu_bwd"; done
echo "== unroll 8, nsplit 16"; BM_HIP_LIB=$R/scripts/micro/bin/libbmhip_u8.so BM_BWD_NSPLIT=16 timeout 120 python scripts/probe_elementwise.py 2>&1 | grep -E "bn_bwd|glu_bwd"
bash scripts/ab_bench.sh ${TAG} "BM_HIP_LIB=$R/scripts/micro/bin/libbmhip_u1.so" "BM_X=0" "BM_HIP_LIB=$R/scripts/micro/bin/libbmhip_u8.so" "BM_BWD_NSPLIT=16" 2>&1 | tail -6
timeout 600 python -m pytest tests -q -m gpu --tb=short -x -k "batchnorm or glu or golden or full_size or gelu" > gpurun_out/${TAG}_tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_tests.log
