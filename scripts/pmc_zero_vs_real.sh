#!/bin/bash
# Same cycles, different clock?  One rocprofv3 counter pass (SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CYCLES, with the
# kernel trace for durations) of a short bench on real operands and one on all-zero operands
# (BM_BENCH_ZERO_OPERANDS=1), condensed per kernel.  Usage (gpurun box): scripts/pmc_zero_vs_real.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-zr}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for V in real zero real2 zero2; do
  D=$R/gpurun_out/prof_${TAG}_$V
  rm -rf $D
  Z=0; case $V in zero*) Z=1;; esac
  BM_BENCH_ZERO_OPERANDS=$Z timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv \
      -d $D -o bench -- python $R/bench.py --steps 3 --warmup 2 \
      --no-cpu-baseline --accuracy-steps 0 --no-exact --no-clip --no-side-blocks > $R/gpurun_out/prof_${TAG}_$V.log 2>&1
  echo "$V rc=$?"
done
cd $R
python - $TAG <<'PY' | tee gpurun_out/${1:-zr}_pmc_zero_vs_real.txt
import collections, csv, sys
from pathlib import Path
tag = sys.argv[1]

def short(name):
    return name.split("(")[0].replace("void ", "").strip()

def load(v):
    root = Path(f"gpurun_out/prof_{tag}_{v}")
    cc = next(root.rglob("*counter_collection.csv"))
    kt = next(root.rglob("*kernel_trace.csv"))
    cnt = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(cc)):
        a = cnt[short(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
    dur = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(kt)):
        a = dur[short(r["Kernel_Name"])]
        a[0] += 1; a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = {}
    for k, c in cnt.items():
        if "SQ_BUSY_CYCLES" not in c or k not in dur:
            continue
        n, s = c["SQ_BUSY_CYCLES"]
        m = c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / max(c["SQ_VALU_MFMA_BUSY_CYCLES"][0], 1)
        out[k] = dict(launches=n, cycles=s / n / 32.0, mfma=m / 1024.0, us=dur[k][1] / dur[k][0] / 1e3)
    return out

runs = {v: load(v) for v in ("real", "zero", "real2", "zero2")}
top = sorted(runs["real"], key=lambda k: -runs["real"][k]["us"] * runs["real"][k]["launches"])[:8]
print("per launch: shader-engine busy cycles (SQ_BUSY_CYCLES / 32), MFMA-busy cycles per SIMD (SQ_VALU_MFMA_BUSY_CYCLES / 1024), "
      "duration in us, clock = cycles / duration")
for k in top:
    print(k[:70])
    for v in ("real", "zero", "real2", "zero2"):
        r = runs[v].get(k)
        if r:
            print(f"   {v:6s} cycles {r['cycles']:10.0f}  mfma-busy {r['mfma']:10.0f} ({r['mfma'] / r['cycles']:.3f})  "
                  f"{r['us']:8.1f} us  {r['cycles'] / r['us'] / 1e3:.3f} GHz")
PY
rm -rf gpurun_out/prof_${TAG}_real gpurun_out/prof_${TAG}_zero gpurun_out/prof_${TAG}_real2 gpurun_out/prof_${TAG}_zero2
