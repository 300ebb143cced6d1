#!/bin/bash
# A/B of switchable features on ONE box (same thermal history): bench.py timed region only, variants interleaved.
# Usage: scripts/ab_bench.sh <tag> "<VAR=VAL ...>" "<VAR=VAL ...>" ...   (first variant is repeated at the end)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
TAG=$1; shift
i=0
run() {
  env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --accuracy-steps 0 --no-exact --no-clip --no-side-blocks \
      > gpurun_out/${TAG}_ab_$2.json 2> gpurun_out/${TAG}_ab_$2.err
  python - "$1" gpurun_out/${TAG}_ab_$2.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = d["roofline"]["per_kernel_ms_per_step"]
    print(f"{sys.argv[1]:55s} ms/step {d['ms_per_step']:.3f}  mfma-kernels {d['roofline']['mfma_kernels_ms_per_step']:.3f}  "
          f"conv<3,5> {k.get('conv_nn_h2w_kernel<3,5>', 0):.3f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
first="$1"
for v in "$@"; do run "$v" $i; i=$((i+1)); done
run "$first" $i
