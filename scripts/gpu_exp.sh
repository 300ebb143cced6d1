#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "== base"; timeout 300 python scripts/probe_conv.py f32x3 2>&1 | grep conv | head -1
for v in NOBARRIER NOX NODMA; do echo "== $v"; BM_HIP_LIB=$PWD/brainmagick_amd/libbmhip_$v.so timeout 300 python scripts/probe_conv.py f32x3 2>&1 | grep conv | head -1; done
