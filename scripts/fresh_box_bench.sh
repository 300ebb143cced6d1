#!/bin/bash
# what the driver does at round end, as the first job of a fresh box: the default bench with its exact command line
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
SECONDS=0
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${1:-fresh}_bench_fresh_box.json 2> gpurun_out/${1:-fresh}_bench_fresh_box.err
echo "rc=$? wall ${SECONDS}s"; cut -c1-330 gpurun_out/${1:-fresh}_bench_fresh_box.json
