"""One-off evidence run (not part of the test-suite: the CPU oracle needs ~1 min per step at this
size): BASELINE cfg2 / cfg3 at the FULL batch of 256 segments -- loss, estimate and every gradient of
the HIP path vs the CPU oracle, written to gpurun_out/fullsize_parity.json."""
import copy
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from brainmagick_amd import synthetic  # noqa: E402
from brainmagick_amd.models import SimpleConv  # noqa: E402
from brainmagick_amd.solver import Solver  # noqa: E402
from oracle import bm_oracle as O  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def main():
    out = {}
    for name in sys.argv[1:] or ["cfg2"]:
        c = synthetic.CONFIGS[name]
        sb = synthetic.make_config_batch(name, seed=2036)
        torch.manual_seed(2036)
        model = SimpleConv(in_channels={"meg": c["C"]}, out_channels=c["F"], hidden={"meg": 320},
                           n_subjects=c["S"], **O.CLIP_CONV_CFG)
        oracle = O.OracleModel(copy.deepcopy(model.state_dict()), O.CLIP_CONV_CFG, 320, c["F"])
        ban = torch.tensor([0.4, 0.6])
        model.merger.ban_center_override = ban
        solver = Solver(model)
        t0 = time.time()
        loss = solver.train_step(sb)
        torch.cuda.synchronize()
        t_hip = time.time() - t0
        t0 = time.time()
        loss_ref, est_ref, grads_ref = oracle.loss_and_grads(sb.meg, sb.positions(), sb.subject_index,
                                                             sb.features, True, ban)
        t_cpu = time.time() - t0
        gscale = max(float(g.norm()) for g in grads_ref.values())
        worst, worst_name = 0.0, ""
        for k, p in model.named_parameters():
            g = grads_ref[k]
            if float(g.abs().max()) <= 1e-5 * gscale:
                continue        # analytically-zero gradient (conv bias before BatchNorm): noise
            # p.grad was consumed by the optimizer step? no: FlatAdam keeps grads until zero_grad
            r = rel(p.grad, g)
            if r > worst:
                worst, worst_name = r, k
        out[name] = dict(batch=c["B"], loss_hip=float(loss), loss_oracle=float(loss_ref),
                         loss_abs_diff=abs(float(loss) - float(loss_ref)),
                         worst_grad_rel_l2=worst, worst_grad=worst_name,
                         hip_first_step_s=t_hip, oracle_step_s=t_cpu,
                         oracle_threads=torch.get_num_threads())
        print(name, out[name], flush=True)
    Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "fullsize_parity.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
