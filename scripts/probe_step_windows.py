"""ms per step over consecutive 20-step windows of one process (does the headline window differ from later ones, and is
the host-side layout de-duplication of fresh segment -> recording assignments the reason?).
    python scripts/probe_step_windows.py [n_draws]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from brainmagick_amd import synthetic  # noqa: E402
from brainmagick_amd.models import SimpleConv  # noqa: E402
from brainmagick_amd.solver import Solver  # noqa: E402

n_draws = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
c = synthetic.CONFIGS["cfg2"]
torch.manual_seed(2036)
model = SimpleConv(in_channels={"meg": c["C"]}, out_channels=c["F"], hidden={"meg": 320}, n_subjects=c["S"],
                   **bench.CLIP_CONV)
solver = Solver(model, device=str(dev))
stream = bench.BatchStream("cfg2", 256, 0, dev, n_draws=n_draws)
for _ in range(5):
    solver.train_step(stream.next()[0])
out = []
for w in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        solver.train_step(stream.next()[0])
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 20 * 1e3)
print(f"n_draws={n_draws}: " + " ".join(f"{t:.2f}" for t in out))
