#!/bin/bash
# round 4, GPU call 1: new tests first (fast feedback), A/B of the side stream / GELU derivative, full suite, default bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
TAG=r4a
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" 2>&1 | tail -1
timeout 900 python -m pytest tests -q -m gpu --tb=short -x \
   -k "copy_stream or writing_grads or negative_pool or per_rank_rejection or whole_node_shape or gelu_and_its or replicas_match or batchnorm_act" \
   > gpurun_out/${TAG}_newtests.log 2>&1
echo "new tests rc=$?"; tail -15 gpurun_out/${TAG}_newtests.log | cut -c1-400
bash scripts/ab_bench.sh ${TAG} "BM_WGRAD_STREAM=0 BM_GELU_GRAD_ERF=1" "BM_WGRAD_STREAM=0" "BM_WGRAD_STREAM=1" "BM_WGRAD_STREAM=1 BM_WGRAD_STREAM_PRIORITY=-1" "BM_WGRAD_STREAM=1 BM_GELU_GRAD_ERF=1" 2>&1 | tail -8
timeout 1500 python -m pytest tests -q -m gpu --tb=short -rf --durations=8 > gpurun_out/${TAG}_tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${TAG}_tests.log | cut -c1-300 | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4a_bench.json').read().strip().splitlines()[-1])
for k in ("overlap","sustained","pcie_inclusive","roofline_retrieval"):
    print(k, json.dumps(d.get(k))[:600])
print("cpu", json.dumps(d.get("cpu_baseline",{}).get("cfg2_b256")))
print("kernels", json.dumps(d["roofline"]["per_kernel_ms_per_step"])[:900])
PY
