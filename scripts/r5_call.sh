#!/bin/bash
# Round-5 development call on a gpurun box: GPU suite (the 20-step full-size horizon test apart), interleaved A/B of
# the switches given as arguments, optionally the default bench.  Usage: scripts/r5_call.sh <tag> <bench:0|1> [A/B variants...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
TAG=$1; BENCH=$2; shift; shift
SECONDS=0
timeout 1200 python -m pytest tests -q -m gpu --tb=short -rf --durations=8 -k "not horizon" > gpurun_out/${TAG}_tests.log 2>&1
echo "pytest rc=$? (${SECONDS}s)"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${TAG}_tests.log | cut -c1-400 | head -30
SECONDS=0
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k horizon -s --tb=short > gpurun_out/${TAG}_horizon.log 2>&1
echo "horizon rc=$? (${SECONDS}s)"; grep -E "full-size horizon|passed|failed|Error|assert" gpurun_out/${TAG}_horizon.log | cut -c1-400 | head -8
if [ $# -gt 0 ]; then bash scripts/ab_bench.sh $TAG "$@"; fi
if [ "$BENCH" = "1" ]; then
  SECONDS=0; timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$? wall ${SECONDS}s"
  python - gpurun_out/${TAG}_bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("ms/step", round(d["ms_per_step"], 3), "seg/s", round(d["value"]), "frac", round(r["frac"], 3), "traffic", r["traffic"], r["traffic_source"],
      "step_hbm_GB", round((r.get("step_hbm_bytes") or 0) / 1e9, 2), "pmc_s", (r.get("traffic_live") or {}).get("seconds"))
print("cpu_baseline", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k in ("value", "cores", "step_s", "kind")})
print("clip", {k: (round(v["kernel_us"], 1), round(v["mfma_frac"], 3), round(v["hbm_frac"], 3)) for k, v in (d.get("roofline_clip") or {}).items() if isinstance(v, dict) and "kernel_us" in v})
print("parity", d.get("retrieval_parity"))
PY
  tail -3 gpurun_out/${TAG}_bench.err
fi
