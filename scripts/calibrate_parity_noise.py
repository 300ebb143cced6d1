"""HIP-only sweep of the planted-noise level of the long-horizon parity run (tests/test_model_gpu.py::PARITY_DIMS):
prints held-out top-1 / top-10 after 200 steps so that PARITY_NOISE can be set away from 0 % and 100 %."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from brainmagick_amd import retrieval, synthetic  # noqa: E402
from brainmagick_amd.losses import ClipLoss  # noqa: E402
from brainmagick_amd.models import SimpleConv  # noqa: E402
from brainmagick_amd.solver import Solver  # noqa: E402
import test_model_gpu as TM  # noqa: E402

d = TM.PARITY_DIMS
for noise in [float(a) for a in sys.argv[1:]] or [2.0, 3.0, 4.0]:
    torch.manual_seed(5)
    model = SimpleConv(in_channels={"meg": d["C"]}, out_channels=d["F"], hidden={"meg": d["hidden"]},
                       n_subjects=d["S"], **TM.parity_model_cfg())
    solver = Solver(model)
    losses = []
    for step in range(200):
        sb = synthetic.make_batch(d["B"], d["C"], d["T"], d["F"], d["S"], seed=100 + step, planted=True, noise=noise)
        losses.append(float(solver.train_step(sb)))
    held = synthetic.make_batch(2048, d["C"], d["T"], d["F"], d["S"], seed=999, planted=True, noise=noise)
    est, cand = solver.predict(held)
    acc = retrieval.segment_topk_accuracy(ClipLoss().cuda(), est, cand, topks=(1, 10))
    print(f"noise {noise}: loss {losses[0]:.3f} -> {sum(losses[-8:]) / 8:.3f}, top-10 {acc['top10']:.4f} top-1 {acc['top1']:.4f}",
          flush=True)
