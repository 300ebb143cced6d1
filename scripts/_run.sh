cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== y_out, publish"; PROBE_OUT=1 timeout 300 python scripts/probe_kernels.py conv f16x2 2>&1 | grep conv
timeout 600 python -m pytest tests/test_f16x2_gpu.py -q --tb=short -x > gpurun_out/t_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|rror" gpurun_out/t_tests.log | cut -c1-250 | head -20
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_t/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --no-cpu-baseline --accuracy-steps 0 --no-exact --no-clip > $GRAFT_REPO_ROOT/gpurun_out/t_bench.json 2>$GRAFT_REPO_ROOT/gpurun_out/t_bench.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,csv,collections
d=json.load(open('gpurun_out/t_bench.json'))
print(d['value'], d['ms_per_step'])
rows=list(csv.DictReader(open('gpurun_out/prof_t/trace/bench_kernel_stats.csv')))
steps=23
for r in rows[:16]:
    print(f"{r['Name'][:60]:60s} {int(r['Calls'])/steps:6.1f}/step {float(r['TotalDurationNs'])/1e6/steps:7.3f} ms/step avg {float(r['AverageNs'])/1e3:8.1f} us")
# per-launch durations of conv<1,5> within one step
tr=list(csv.DictReader(open('gpurun_out/prof_t/trace/bench_kernel_trace.csv')))
k=[(int(r['Start_Timestamp']), r['Kernel_Name'][:40], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in tr]
k.sort()
sel=[x for x in k if 'conv_nn_h2w_kernel<1, 5>' in x[1]]
print([round(x[2]) for x in sel[-18:]])
PY
find gpurun_out/prof_t -name "*kernel_trace.csv" -delete
