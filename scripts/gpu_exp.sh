#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "== base"; timeout 300 python scripts/probe_gemm.py f32x3 2>&1 | grep wgrad | head -3 | tail -1
for v in NOSPLIT NOLOAD NOBAR; do echo "== $v"; BM_HIP_LIB=$PWD/brainmagick_amd/libbmhip_$v.so timeout 300 python scripts/probe_gemm.py f32x3 2>&1 | grep wgrad | head -3 | tail -1; done
