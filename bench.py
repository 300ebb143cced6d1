#!/usr/bin/env python
"""Headline benchmark of the MI355X-native SimpleConv + ClipLoss training step.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = the whole hot path on one synthetic batch already resident in HBM
(bm/solver.py:343-390): SimpleConv forward, ClipLoss, backward, gradient exchange, fused Adam.
Workload = BASELINE.json configs[1] (gwilliams2022-shaped MEG: 208 sensors x 360 samples, 120 mel
features, 27 subjects, batch 256 per GPU, the paper's clip_conv model).  With N > 1 every rank
processes its own 256 segments (weak scaling), candidates are all-gathered so the negatives pool is
whole-node (configs[3]) and gradients go through ONE reduce-scatter + all-gather on the flat bucket.

Rank 0 prints ONE JSON line with the driver's contract plus ``roofline`` (dominant kernel, timed
live with HIP events on the launch stream inside the timed region) and ``cpu_baseline`` (the CPU
oracle = torch-CPU restatement of the reference, timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from brainmagick_amd import distrib, hip_ops, synthetic  # noqa: E402
from brainmagick_amd.models import SimpleConv  # noqa: E402
from brainmagick_amd.solver import Solver  # noqa: E402

# conf/model/clip_conv.yaml:5-38 (the paper model) -- kept literal here: the product does not
# import the oracle.
CLIP_CONV = dict(depth=10, kernel_size=3, dilation_growth=2, dilation_period=5, batch_norm=True,
                 skip=True, gelu=True, glu=2, glu_context=1, glu_glu=True, complex_out=True,
                 merger=True, merger_pos_dim=2048, merger_channels=270, merger_dropout=0.2,
                 merger_penalty=0., initial_linear=270, initial_depth=1, subject_layers=True,
                 subject_layers_dim="input", subject_dim=0)

PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense bf16 MFMA peak (opt-in --dtype bf16 mode only)
PEAK_HBM_GBS = 8000.0


def algorithmic_flops_per_segment(C, T, F, hidden=320, merger_ch=270) -> float:
    """SURVEY.md §8d (de-duplicated merger scores): forward MACs per segment, x2 FLOP, x3 fwd+bwd."""
    mac = merger_ch * C * T                       # merger apply
    mac += merger_ch * merger_ch * T * 2          # initial_linear + subject layer
    cin = merger_ch
    for k in range(10):
        mac += hidden * cin * 3 * T
        cin = hidden
        if k % 2 == 1:
            mac += 2 * hidden * hidden * 3 * T    # GLU conv
    mac += 2 * hidden * hidden * T + 2 * hidden * F * T   # head
    return 2.0 * 3.0 * mac


def pmc_traffic(kernel_label: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (FETCH_SIZE / WRITE_SIZE collected in separate passes and corrected per MI355X_MICROARCH.md;
    see profiles/*pmc_summary.json).  PMC counters cannot be sampled from inside this process."""
    files = sorted((ROOT / "profiles").glob("*pmc_summary*.json"))
    if not files:
        return None, None
    data = json.loads(files[-1].read_text())
    prefix = kernel_label.rstrip(">")
    for name, rec in data.get("kernels", {}).items():
        if name.startswith(prefix):
            return rec.get("hbm_bytes_per_launch_corrected"), files[-1].name
    return None, files[-1].name


def cpu_baseline(workload, seconds_budget=20.0):
    """The CPU oracle (port of the reference path) timed on the host cores, bounded sample."""
    from oracle import bm_oracle as O
    B = 16
    c = synthetic.CONFIGS[workload]
    sb = synthetic.make_batch(B, c["C"], c["T"], c["F"], c["S"], seed=2036)
    torch.manual_seed(0)
    model = SimpleConv(in_channels={"meg": c["C"]}, out_channels=c["F"], hidden={"meg": 320},
                       n_subjects=c["S"], **CLIP_CONV)
    oracle = O.OracleModel(model.state_dict(), O.CLIP_CONV_CFG, 320, c["F"])
    ban = torch.tensor([0.5, 0.5])
    pos = sb.positions()
    times = []
    t_start = time.perf_counter()
    steps = 0
    while steps < 2 or (time.perf_counter() - t_start < seconds_budget and steps < 12):
        t0 = time.perf_counter()
        oracle.train_step(sb.meg, pos, sb.subject_index, sb.features, ban)
        times.append(time.perf_counter() - t0)
        steps += 1
    timed = sorted(times[1:])
    med = timed[len(timed) // 2]
    return dict(value=B / med, unit="segments/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{len(timed)} train steps of batch {B} ({workload} shapes), median step "
                       f"{med * 1e3:.0f} ms, torch CPU fp32")


def retrieval_block(workload, B, steps, dev, n_train=8, n_held=4):
    """top-k segment retrieval (scripts/run_eval_probs.py:237-264 rule) of the full-size model after `steps`
    training steps on planted-latent synthetic batches, evaluated on held-out segments of the
    same synthetic world.  Chance level for top-10 is 10 / (n_held * B)."""
    from brainmagick_amd import retrieval
    from brainmagick_amd.losses import ClipLoss
    c = synthetic.CONFIGS[workload]
    kw = dict(planted=True)
    if workload == "cfg5":
        kw["mixed_eeg"] = True
    torch.manual_seed(77)
    model = SimpleConv(in_channels={"meg": c["C"]}, out_channels=c["F"], hidden={"meg": 320},
                       n_subjects=c["S"], **CLIP_CONV)
    solver = Solver(model, device=str(dev), negatives="local")
    train = [synthetic.make_batch(B, c["C"], c["T"], c["F"], c["S"], seed=500 + i, **kw).to(dev)
             for i in range(n_train)]
    losses = [float(solver.train_step(train[i % n_train])) for i in range(steps)]
    ests, cands = [], []
    for i in range(n_held):
        held = synthetic.make_batch(B, c["C"], c["T"], c["F"], c["S"], seed=9000 + i, **kw).to(dev)
        e, o = solver.predict(held)
        ests.append(e)
        cands.append(o)
    acc = retrieval.segment_topk_accuracy(ClipLoss().to(dev), torch.cat(ests), torch.cat(cands), topks=(1, 10))
    n = n_held * B
    return {"top1": acc["top1"], "top10": acc["top10"], "chance_top10": 10.0 / n, "held_out_segments": n,
            "train_steps": steps, "train_batches": n_train, "first_loss": losses[0], "last_loss": losses[-1],
            "data": "planted-latent synthetic world (brainmagick_amd/synthetic.py), default compute mode"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg5"])
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--negatives", default=None, choices=["local", "node"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--accuracy-steps", type=int, default=60,
                    help="second half of the metric: after the timed region (never part of `value`), train a fresh "
                         "model this many steps on planted-signal synthetic batches and report top-1/top-10 "
                         "segment retrieval on held-out segments (single-GPU runs only; 0 disables)")
    ap.add_argument("--no-exact", action="store_true",
                    help="skip the extra exact-fp32 MFMA timing block (profiling runs)")
    ap.add_argument("--dtype", default="f32x3", choices=["f32x3", "f32", "bf16"],
                    help="compute mode of the contractions. f32x3 (default): fp32-accurate 3xbf16-split "
                         "emulation on the bf16 matrix cores (same parity tolerances as exact fp32); f32: "
                         "exact-fp32 MFMA; bf16: opt-in REDUCED precision, for information only")
    args = ap.parse_args()

    hip_ops.set_compute_dtype(args.dtype)
    distrib.init()
    rank, world = distrib.rank(), distrib.world_size()
    if world != args.gpus:
        if args.gpus != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    c = synthetic.CONFIGS[args.workload]
    B = args.batch
    negatives = args.negatives or ("node" if world > 1 else "local")
    torch.manual_seed(2036)
    model = SimpleConv(in_channels={"meg": c["C"]}, out_channels=c["F"], hidden={"meg": 320},
                       n_subjects=c["S"], **CLIP_CONV)
    n_params = sum(p.numel() for p in model.parameters())
    solver = Solver(model, device=str(dev), negatives=negatives)
    batch = synthetic.make_config_batch(args.workload, seed=2036 + rank, batch=B).to(dev)

    for _ in range(args.warmup):
        solver.train_step(batch)
    timer = hip_ops.KernelTimer()
    distrib.barrier()
    torch.cuda.synchronize()
    hip_ops.set_kernel_timer(timer)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = solver.train_step(batch)
    distrib.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    hip_ops.set_kernel_timer(None)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss)

    # For transparency the same job is also timed in the exact-fp32 MFMA mode (not part of `value`).
    exact = None
    if args.dtype == "f32x3" and not args.no_exact:
        hip_ops.set_compute_dtype("f32")
        k2 = max(2, args.steps // 4)
        solver.train_step(batch)
        distrib.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k2):
            solver.train_step(batch)
        distrib.barrier()
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([e2], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            e2 = float(t)
        exact = {"compute_mode": "f32 (exact-fp32 MFMA v_mfma_f32_32x32x2_f32)", "steps": k2,
                 "ms_per_step": e2 / k2 * 1e3, "value": world * B * k2 / e2, "unit": "segments/s"}
        hip_ops.set_compute_dtype(args.dtype)

    if rank != 0:
        return
    ksum = timer.summary()
    dom = max(ksum, key=lambda k: ksum[k]["avg_ms"] * ksum[k]["launches"])
    d = ksum[dom]
    achieved = d["flops_per_launch"] / (d["avg_ms"] * 1e-3) / 1e12
    seg_flops = algorithmic_flops_per_segment(c["C"], c["T"], c["F"])
    traffic, traffic_src = pmc_traffic(dom)
    # f32: exact-fp32 MFMA peak.  f32x3: six bf16 MFMAs per fp32-accurate block -> the algorithmic
    # (fp32-equivalent) FLOP/s are priced against 1/6 of the dense bf16 peak.  bf16: dense bf16 peak.
    peak_tf = {"f32": PEAK_FP32_MFMA_TFLOPS, "f32x3": PEAK_BF16_MFMA_TFLOPS / 6.0,
               "bf16": PEAK_BF16_MFMA_TFLOPS}[args.dtype]
    total_kernel_ms = sum(v["avg_ms"] * v["launches"] for v in ksum.values()) / args.steps
    out = {
        "metric": "segments/s, 208-ch x 360-sample SimpleConv + ClipLoss training step",
        "value": world * B * args.steps / elapsed,
        "unit": "segments/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32x3": "f32 (fp32-accurate: operands split exactly into 3 bf16 planes, 6 MFMA products, "
                           "fp32 accumulate; error vs fp64 <= exact-fp32 MFMA, see DESIGN.md §2)",
                  "f32": "f32 (exact-fp32 MFMA)", "bf16": "bf16 operands, f32 accumulate (REDUCED "
                                                          "precision, information only)"}[args.dtype],
        "compute_mode": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.workload}: gwilliams2022-shaped MEG C={c['C']} T={c['T']}, "
                               f"F={c['F']} features, {c['S']} subjects, clip_conv SimpleConv "
                               f"({n_params} params) + ClipLoss + Adam",
                   "batch_per_gpu": B, "global_batch": world * B, "negatives": negatives,
                   "parallelism": f"dp{world}", "final_loss": final_loss,
                   "step_tflops": seg_flops * B * world * args.steps / elapsed / 1e12},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved,
                     "peak": peak_tf, "unit": "TFLOP/s", "frac": achieved / peak_tf,
                     "traffic": traffic,
                     "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, separate passes)",
                     "traffic_source": traffic_src,
                     "avg_launch_ms": d["avg_ms"], "launches_per_step": d["launches"] / args.steps,
                     "algorithmic_flops_per_launch": d["flops_per_launch"],
                     "mfma_kernels_ms_per_step": total_kernel_ms,
                     "hbm_roofline_frac_step": (B * 69.2e6 + 108e6) / (PEAK_HBM_GBS * 1e9)
                     / (elapsed / args.steps) if args.workload == "cfg2" else None},
    }
    out["exact_f32_mfma"] = exact
    out["retrieval"] = None
    if world == 1 and args.accuracy_steps > 0:
        try:
            out["retrieval"] = retrieval_block(args.workload, B, args.accuracy_steps, dev)
        except Exception as exc:   # the throughput line must survive a failure of the side measurement
            out["retrieval"] = {"error": repr(exc)}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


if __name__ == "__main__":
    try:
        main()
    finally:
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
