"""PCIe-inclusive step rate: the same cfg2 training step with the batch handed over as HOST buffers
(bm/solver.py:236 `batch.to(device)`), pageable and pinned.  Never the bench `value` (inputs resident in HBM)."""
import json
import sys
import time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from brainmagick_amd import synthetic
from brainmagick_amd.models import SimpleConv
from brainmagick_amd.solver import Solver
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench

c = synthetic.CONFIGS["cfg2"]
torch.manual_seed(2036)
model = SimpleConv(in_channels={"meg": c["C"]}, out_channels=c["F"], hidden={"meg": 320}, n_subjects=c["S"],
                   **bench.CLIP_CONV)
solver = Solver(model, device="cuda")
host = synthetic.make_config_batch("cfg2", seed=2036, batch=256)
res = {}
for tag in ("resident", "pageable", "pinned"):
    b = host
    if tag == "pinned":
        b = synthetic.SegmentBatch(host.meg.pin_memory(), host.features.pin_memory(), host.features_mask.pin_memory(),
                                   host.subject_index.pin_memory(), host.recording_index.pin_memory(), host._recordings)
    dev_b = host.to("cuda")
    for _ in range(3):
        solver.train_step(dev_b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        solver.train_step(dev_b if tag == "resident" else b.to("cuda"))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    res[tag] = {"ms_per_step": dt * 1e3, "segments_per_s": 256 / dt}
res["bytes_per_batch"] = int(host.meg.numel() * 4 + host.features.numel() * 4)
print(json.dumps(res))
