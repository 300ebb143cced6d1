#!/bin/bash
# Side benches of the evidence set alone (no suite): cfg3, cfg5, 400 steps sustained, the RCCL path forced at world 1 with
# whole-node negatives (prints comm_ms), real / all-zero operands interleaved.  Usage: scripts/r5_side.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
TAG=${1:-r5e}
SIDE="--no-cpu-baseline --accuracy-steps 0 --no-exact --no-clip --no-side-blocks"
timeout 300 python bench.py $SIDE > gpurun_out/${TAG}_real.json 2>/dev/null
timeout 300 python bench.py --workload cfg3 $SIDE > gpurun_out/${TAG}_bench_cfg3.json 2>/dev/null; echo "cfg3 rc=$?"
timeout 300 python bench.py --workload cfg5 $SIDE > gpurun_out/${TAG}_bench_cfg5.json 2>/dev/null; echo "cfg5 rc=$?"
timeout 300 python bench.py --steps 400 --warmup 5 $SIDE > gpurun_out/${TAG}_bench_sustained_400steps.json 2>/dev/null; echo "sustained rc=$?"
BM_FORCE_DISTRIBUTED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29000 + RANDOM % 900)) timeout 300 \
  python bench.py --negatives node $SIDE > gpurun_out/${TAG}_bench_forced_rccl_world1_node.json 2> gpurun_out/${TAG}_bench_forced.err; echo "forced rccl rc=$?"
BM_BENCH_ZERO_OPERANDS=1 timeout 300 python bench.py $SIDE > gpurun_out/${TAG}_zero.json 2>/dev/null
timeout 300 python bench.py $SIDE > gpurun_out/${TAG}_real2.json 2>/dev/null
for f in real bench_cfg3 bench_cfg5 bench_sustained_400steps bench_forced_rccl_world1_node zero real2; do
  python - gpurun_out/${TAG}_$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "ms/step", round(d["ms_per_step"], 3), "seg/s", round(d["value"]), d["config"].get("comm"), d.get("comm_ms") and {k: round(v["ms_per_step"], 3) for k, v in d["comm_ms"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
