"""Times the CPU oracle (`oracle/bm_oracle.py`, the "port" behind bench.py's cpu_baseline) BESIDE THE LIVE REFERENCE
(`/root/reference/bm/models/simpleconv.py` + `bm/losses.py` + torch.optim.Adam, imported under the stubs of
tests/golden/_ref_import.py) on the same batch, the same initial state and the same thread count, interleaved, in the
build container -- the GPU box has no /root/reference, so `cpu_baseline.kind` stays "port" there and this file is
what backs it:  python scripts/port_vs_reference.py > profiles/r5_port_vs_reference.json

Workloads: cfg1 = BASELINE configs[0] (C=273 T=360 F=120, batch 16) in full, and cfg2's shapes at batch 32 (the
container has 8 cores and cannot hold batch 256 in a sensible time).  Reported: segments/s of both, their ratio,
and the loss both paths print for the same step (they agree to fp32 round-off: the port IS the reference's arithmetic).
"""
import copy
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "golden"))

from _ref_import import load_reference  # noqa: E402
from brainmagick_amd import synthetic  # noqa: E402
from oracle import bm_oracle as O  # noqa: E402


class _Batch:
    def __init__(self, sb):
        self.meg, self.subject_index, self._recordings = sb.meg, sb.subject_index, sb._recordings
        self._positions = sb.positions()

    def __len__(self):
        return len(self.meg)


def one_config(name, B, steps, threads):
    sc, common, losses = load_reference()
    c = synthetic.CONFIGS[name]
    sb = synthetic.make_config_batch(name, seed=2036, batch=B)
    torch.manual_seed(0)
    ref = sc.SimpleConv(in_channels={"meg": c["C"]}, out_channels=c["F"], hidden={"meg": 320}, n_subjects=c["S"],
                        **O.CLIP_CONV_CFG)
    ref.train(True)
    sd0 = copy.deepcopy(ref.state_dict())
    oracle = O.OracleModel(copy.deepcopy(sd0), O.CLIP_CONV_CFG, 320, c["F"])
    loss_mod = losses.ClipLoss()
    optim = torch.optim.Adam(ref.parameters(), lr=3e-4, betas=(0.9, 0.999))
    batch = _Batch(sb)
    ban = torch.tensor([0.5, 0.5])
    common.PositionGetter.get_positions = lambda self, b: b._positions.clone()
    real_rand = torch.rand
    common.torch.rand = lambda *a, **k: ban.clone() if a == (2,) else real_rand(*a, **k)
    mask = torch.ones(B, 1, c["T"], dtype=torch.bool)
    pos = sb.positions()
    torch.set_num_threads(threads)

    def ref_step():
        est = ref({"meg": sb.meg.clone()}, batch)
        loss = loss_mod(est, sb.features, mask)
        optim.zero_grad()
        loss.backward()
        optim.step()
        return float(loss)

    def port_step():
        return float(oracle.train_step(sb.meg, pos, sb.subject_index, sb.features, ban)[0])

    t_ref, t_port, l_ref, l_port = [], [], [], []
    try:
        for i in range(steps + 1):              # interleaved; the first pair is the warm-up
            t0 = time.perf_counter(); l_ref.append(ref_step()); t1 = time.perf_counter()
            l_port.append(port_step()); t2 = time.perf_counter()
            if i:
                t_ref.append(t1 - t0)
                t_port.append(t2 - t1)
    finally:
        common.torch.rand = real_rand
    med = lambda v: sorted(v)[len(v) // 2]       # noqa: E731
    return {"config": f"{name} shapes (C={c['C']} T={c['T']} F={c['F']}), batch {B}, clip_conv model, whole training "
                      "step (forward + ClipLoss + backward + Adam)", "threads": threads, "steps_timed": steps,
            "reference_step_s": med(t_ref), "port_step_s": med(t_port),
            "reference_seg_per_s": B / med(t_ref), "port_seg_per_s": B / med(t_port),
            "port_over_reference": med(t_ref) / med(t_port),
            "losses_reference": l_ref, "losses_port": l_port,
            "max_abs_loss_gap": max(abs(a - b) for a, b in zip(l_ref, l_port))}


if __name__ == "__main__":
    threads = min(8, os.cpu_count() or 1)
    cfg1 = one_config("cfg1", 16, 8, threads)
    cfg2 = one_config("cfg2", 32, 6, threads)
    out = dict(cfg1, cfg2_b32=cfg2,
               where=f"build container ({os.cpu_count()} cores), torch {torch.__version__}, live reference from "
                     "/root/reference imported under the stubs of tests/golden/_ref_import.py",
               note="port = oracle/bm_oracle.py (what bench.py's cpu_baseline times on the GPU box's host); a ratio of "
                    "~1 means the port costs what the reference costs")
    print(json.dumps(out, indent=1))
