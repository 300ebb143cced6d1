#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
python - <<'PY' 2>&1 | tail -30
import torch, bench
from brainmagick_amd import hip_ops as H, synthetic
from brainmagick_amd.models import SimpleConv
from brainmagick_amd.solver import Solver
c = synthetic.CONFIGS["cfg2"]
model = SimpleConv(in_channels={"meg": c["C"]}, out_channels=c["F"], hidden={"meg": 320}, n_subjects=c["S"], **bench.CLIP_CONV)
solver = Solver(model, device="cuda:0")
stream = bench.BatchStream("cfg2", 256, 0, torch.device("cuda:0"))
for _ in range(2): solver.train_step(stream.next())
orig = H._PackPlan.get
def get(self, src, geom):
    key = (src.data_ptr(),) + geom
    e = self.entries.get(key)
    if e is None:
        print("NEW", geom, src.shape, src.requires_grad, src.is_leaf)
    elif e["stamp"] != (H._weights_epoch, src._version):
        print("STALE", geom[:4], "stamp", e["stamp"], "now", (H._weights_epoch, src._version), "reg ver", e["src"]._version, src is e["src"])
    return orig(self, src, geom)
H._PackPlan.get = get
for _ in range(2):
    print("--- step")
    solver.train_step(stream.next())
PY
