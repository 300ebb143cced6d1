#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
TAG=r4e
timeout 900 python -m pytest tests -q -m gpu --tb=short -k "against_reference_golden or deferred_asserts or staged_on_the_copy or state_dict or unsupported" > gpurun_out/${TAG}_newtests.log 2>&1
echo "new tests rc=$?"; tail -25 gpurun_out/${TAG}_newtests.log | cut -c1-400
SIDE="--no-cpu-baseline --accuracy-steps 0 --no-exact --no-clip --no-side-blocks"
for i in 0 1; do
timeout 300 python bench.py $SIDE > gpurun_out/${TAG}_real_$i.json 2> gpurun_out/${TAG}_real_$i.err; echo "real rc=$?"
BM_BENCH_ZERO_OPERANDS=1 timeout 300 python bench.py $SIDE > gpurun_out/${TAG}_zero_$i.json 2> gpurun_out/${TAG}_zero_$i.err; echo "zero rc=$?"
done
python - <<'PY'
import json
for n in ("real_0","zero_0","real_1","zero_1"):
    try:
        d=json.loads(open(f"gpurun_out/r4e_{n}.json").read().strip().splitlines()[-1])
        k=d["roofline"]["per_kernel_ms_per_step"]
        print(n, "ms/step", round(d["ms_per_step"],3), "event pass", round(d["roofline"]["event_pass_ms_per_step"],3), "loss", d["config"]["final_loss"], {a:round(b,3) for a,b in list(k.items())[:4]})
    except Exception as e: print(n,"ERR",e)
PY
