#!/bin/bash
# Diagnostic build: libbmhip with the in-kernel cycle trace of the wide kernels compiled in (${BM_TRACE_DEFS:--DHG_TRACE}).
# Use with BM_HIP_LIB=brainmagick_amd/libbmhip_trace.so.
set -e
cd "$(dirname "$0")/../brainmagick_amd/csrc"
O=$(mktemp -d)
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc ${BM_TRACE_DEFS:--DHG_TRACE} -c $f -o $O/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -shared $O/*.o -o ../${BM_TRACE_OUT:-libbmhip_trace.so}
rm -rf $O
echo built ../${BM_TRACE_OUT:-libbmhip_trace.so}
