#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
TAG=r4b
timeout 900 python -m pytest tests -q -m gpu --tb=short \
   -k "copy_stream or writing_grads or negative_pool or per_rank_rejection or whole_node_shape or replicas_match or grouped_weight or against_reference_golden or accept_per_rank or training_curve" \
   > gpurun_out/${TAG}_newtests.log 2>&1
echo "new tests rc=$?"; tail -12 gpurun_out/${TAG}_newtests.log | cut -c1-300
bash scripts/ab_bench.sh ${TAG} "BM_WGRAD_STREAM=0" "BM_WGRAD_STREAM=1" "BM_WGRAD_STREAM=1 BM_WGRAD_STREAM_PRIORITY=-1" "BM_WGRAD_STREAM=0 BM_GELU_GRAD_ERF=1" "BM_WGRAD_STREAM=0 BM_H2_GROUPED=0" "BM_WGRAD_STREAM=0 BM_SWAP_TRANSPOSED_WGRAD=0" 2>&1 | tail -8
for i in 0 1 2 3 4 5 6; do python - $i <<'PY'
import json,sys
i=sys.argv[1]
try:
    d=json.load(open(f'gpurun_out/r4b_ab_{i}.json'))
    k=d['roofline']['per_kernel_ms_per_step']
    print(i, round(d['ms_per_step'],3), 'loss', d['config']['final_loss'], 'single', round(d['overlap']['single_stream_ms_per_step'],3), 'evpass', round(d['roofline']['event_pass_ms_per_step'],3), {a:round(b,3) for a,b in list(k.items())[:7]})
except Exception as e: print(i,'ERR',e)
PY
done
timeout 1500 python -m pytest tests -q -m gpu --tb=short -rf --durations=8 > gpurun_out/${TAG}_tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${TAG}_tests.log | cut -c1-300 | head -20
