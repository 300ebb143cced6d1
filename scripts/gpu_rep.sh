#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/scripts
for lib in libbmhip.so libbmhip_rep6.so; do
  echo == $lib
  BM_HIP_LIB=$R/brainmagick_amd/$lib timeout 600 python perf_probe_bf16.py 2>&1 | grep "^bf16"
done
