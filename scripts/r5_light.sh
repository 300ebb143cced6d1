#!/bin/bash
# Lighter evidence call: the 20-step full-size horizon test, smoke, the timed default bench, rocprofv3 stats + PMC passes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
TAG=${1:-r5b}
SECONDS=0
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k horizon -s --tb=short > gpurun_out/${TAG}_horizon.log 2>&1
echo "horizon rc=$? (${SECONDS}s)"; grep -E "full-size horizon|passed|failed|Error|assert" gpurun_out/${TAG}_horizon.log | cut -c1-300 | head -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SECONDS=0; timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$? wall ${SECONDS}s"
python - gpurun_out/${TAG}_bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("ms/step", round(d["ms_per_step"], 3), "seg/s", round(d["value"]), "frac", round(r["frac"], 3), "traffic", r["traffic"], r["traffic_source"],
      "step_hbm_GB", round((r.get("step_hbm_bytes") or 0) / 1e9, 2), "zero-operand", (d.get("dvfs_probe") or {}).get("zero_operand_ms_per_step"))
PY
bash scripts/gpu_profile.sh $TAG 2>&1 | tail -10
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | head -4
