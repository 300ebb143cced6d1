#!/bin/bash
# One GPU-box call that produces the round's committed evidence: full GPU suite, smoke, default bench (timed), rocprofv3
# kernel stats + PMC passes, the side benches (cfg3, cfg5, sustained, RCCL path forced at world size 1 with whole-node
# negatives + prefetch), the zero-operand DVFS probe and a power / clock trace.  Usage: scripts/final_evidence.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
TAG=${1:-final}
bash scripts/gpu_check.sh $TAG 1500 prof 2>&1 | tail -30
cp gpurun_out/${TAG}_pmc_summary.json gpurun_out/${TAG}_pmc_summary_onbox.json 2>/dev/null
SIDE="--no-cpu-baseline --accuracy-steps 0 --no-exact --no-clip --no-side-blocks"
timeout 300 python bench.py --workload cfg3 $SIDE > gpurun_out/${TAG}_bench_cfg3.json 2>/dev/null; echo "cfg3 rc=$?"
timeout 300 python bench.py --workload cfg5 $SIDE > gpurun_out/${TAG}_bench_cfg5.json 2>/dev/null; echo "cfg5 rc=$?"
# power / clock trace next to a 400-step run (rocm-smi sampled every 100 ms; read-only)
( for i in $(seq 1 70); do rocm-smi --showpower --showclocks --csv 2>/dev/null | tr '\n' ' '; echo; sleep 0.1; done > gpurun_out/${TAG}_power_trace.txt ) &
timeout 300 python bench.py --steps 400 --warmup 5 $SIDE > gpurun_out/${TAG}_bench_sustained_400steps.json 2>/dev/null; echo "sustained rc=$?"
wait
rocm-smi --showmaxpower 2>/dev/null | grep -i -E "power|watt" | head -3
BM_FORCE_DISTRIBUTED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29000 + RANDOM % 900)) timeout 300 \
  python bench.py --negatives node $SIDE > gpurun_out/${TAG}_bench_forced_rccl_world1_node.json 2> gpurun_out/${TAG}_bench_forced.err; echo "forced rccl rc=$?"
timeout 300 python bench.py $SIDE > gpurun_out/${TAG}_real.json 2>/dev/null
BM_BENCH_ZERO_OPERANDS=1 timeout 300 python bench.py $SIDE > gpurun_out/${TAG}_zero.json 2>/dev/null
for f in bench_cfg3 bench_cfg5 bench_sustained_400steps bench_forced_rccl_world1_node real zero; do
  python - gpurun_out/${TAG}_$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "ms/step", round(d["ms_per_step"], 3), "seg/s", round(d["value"]), d["config"].get("comm"), d["config"].get("negatives"), d["config"].get("candidate_gather"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
python scripts/probe_bn_bwd.py > gpurun_out/${TAG}_bn_bwd_probe.txt 2>&1; cat gpurun_out/${TAG}_bn_bwd_probe.txt | tail -3
head -3 gpurun_out/${TAG}_power_trace.txt | cut -c1-300
