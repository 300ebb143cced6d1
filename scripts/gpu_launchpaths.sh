#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "== torchrun nproc 1"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{"metric' | cut -c1-200
echo "== forced distributed (RCCL collectives at world 1, node negatives)"
BM_FORCE_DISTRIBUTED=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29562 timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --negatives node > gpurun_out/forced.log 2>&1
grep '^{"metric' gpurun_out/forced.log | cut -c1-330; grep -v '^{"metric' gpurun_out/forced.log | tail -5
echo "== default bench wall time"
time (timeout 900 python bench.py 2>&1 | tail -1 | cut -c1-160)
