#!/usr/bin/env python
"""Headline benchmark of the MI355X-native SimpleConv + ClipLoss training step.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment the script re-launches itself as
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py``
(one rank per GPU); launched under torchrun by the driver it uses the environment it finds.

One "step" = the whole hot path on one synthetic batch already resident in HBM
(bm/solver.py:343-390): SimpleConv forward, ClipLoss, backward, gradient exchange, fused Adam.
Workload = BASELINE.json configs[1] (gwilliams2022-shaped MEG: 208 sensors x 360 samples, 120 mel
features, 27 subjects, batch 256 per GPU, the paper's clip_conv model).  The timed loop cycles through
8 distinct batches per rank and draws a fresh segment -> recording assignment every step, so the
per-step host work of a shuffled training stream (layout de-duplication, index upload) is inside the
timed region.  With N > 1 every rank processes its own 256 segments (weak scaling), candidates are
all-gathered so the negatives pool is whole-node (configs[3]) and gradients go through ONE in-place
reduce-scatter + all-gather on the flat bucket, on RCCL behind the C-ABI (bm_comm_*).

Rank 0 prints ONE JSON line with the driver's contract plus
  ``roofline``       dominant kernel, timed with HIP events on the launch stream in a second, untimed
                     pass over the same batches right after the timed region (no event records inside
                     the timed region); ``traffic`` = HBM bytes per launch MEASURED IN THIS RUN by two
                     ``rocprofv3 --pmc`` children (FETCH_SIZE, WRITE_SIZE) over a 5-step child of this script
                     while this process leaves the GPU alone (``traffic_source`` "live"; the committed
                     ``profiles/*pmc_summary.json`` is the fallback: "committed:<file>");
  ``roofline_clip``  the ClipLoss contraction (bm/losses.py:91-95) the same way, for cfg2 - cfg4 shapes;
  ``comm_ms``        (N > 1) HIP-event time per step of every collective, per phase;
  ``cpu_baseline``   the CPU oracle (torch-CPU restatement of the reference path, kind "port": the
                     reference modules themselves need /root/reference, absent on the GPU box; timed beside
                     the live reference in the build container: profiles/r5_port_vs_reference.json) on the
                     configuration of ``value`` itself (cfg2, batch 256, 32 threads), with cfg1 (batch 16,
                     1 thread = what bm/train.py:182 configures, and 32 threads) as a sub-block.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from brainmagick_amd import _lib, distrib, hip_ops, synthetic  # noqa: E402
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
PEAK_16BIT_MFMA_TFLOPS = 2500.0     # dense bf16 / f16 MFMA peak
PEAK_HBM_GBS = 8000.0
# MFMA products per fp32-accurate block of each compute mode -> fp32-equivalent peak = 16-bit peak / products
MODE_PEAK_TFLOPS = {"f32": PEAK_FP32_MFMA_TFLOPS, "f32x3": PEAK_16BIT_MFMA_TFLOPS / 6.0,
                    "f16x2": PEAK_16BIT_MFMA_TFLOPS / 3.0}
MODE_DTYPE = {
    "f16x2": "f32 (fp32-accurate: operands split into 2 scaled f16 planes, 3 MFMA products, fp32 accumulate; "
             "same parity tolerances as exact fp32, see DESIGN.md §2)",
    "f32x3": "f32 (fp32-accurate: operands split exactly into 3 bf16 planes, 6 MFMA products, "
             "fp32 accumulate; error vs fp64 <= exact-fp32 MFMA, see DESIGN.md §2)",
    "f32": "f32 (exact-fp32 MFMA)"}


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


def algorithmic_bytes_per_step(C, T, F, B, n_params, hidden=320, merger_ch=270) -> float:
    """4*(C*T + 5*A) bytes per segment + 12 bytes per parameter per step."""
    A = merger_ch * T * 3 + hidden * T * 10 + 2 * hidden * T * 5 + hidden * T * 5 + 2 * hidden * T + F * T
    return B * 4.0 * (C * T + 5 * A) + 12.0 * n_params


def pmc_traffic(kernel_label: str):
    """(HBM bytes per launch, summary file, that summary's average launch duration in ms) of the dominant kernel from
    the newest committed rocprofv3 PMC summary (FETCH_SIZE / WRITE_SIZE collected in separate passes and corrected
    per MI355X_MICROARCH.md; see profiles/*pmc_summary.json).  PMC counters cannot be sampled from inside this
    process; the caller compares the summary's launch duration with the one it has just measured and marks the
    figure stale when the kernel has changed since the counters were collected."""
    files = sorted((ROOT / "profiles").glob("*pmc_summary*.json"))      # rNx_ tags: the last name is the newest round
    if not files:
        return None, None, None
    data = json.loads(files[-1].read_text())
    prefix = kernel_label.rstrip(">").replace(" ", "")
    nb = nt = nn = 0.0
    for name, rec in data.get("kernels", {}).items():       # launch-weighted over the label's instantiations
        if name.replace(" ", "").replace("void", "", 1).startswith(prefix) and rec.get("hbm_bytes_per_launch_corrected"):
            n = float(rec.get("launches") or 1)
            nb += rec["hbm_bytes_per_launch_corrected"] * n
            nt += (rec.get("avg_ns") or 0.0) * n
            nn += n
    if nn:
        return nb / nn, files[-1].name, (nt / nn * 1e-6 if nt else None)
    return None, files[-1].name, None


def _pmc_counter_csv(counter, args, timeout):
    """One `rocprofv3 --pmc <counter> --kernel-trace` pass over a 2-step child of this script (cwd /tmp, TMPDIR=/tmp,
    counters in their own run: MI355X_MICROARCH.md, rocprofv3 section).  Returns {kernel: [launches, sum]} or None."""
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix="bm_pmc_", dir="/tmp")
    cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "bench", "--",
           sys.executable, str(Path(__file__).resolve()), "--steps", "2", "--warmup", "1", "--workload", args.workload,
           "--batch", str(args.batch), "--dtype", args.dtype, "--no-cpu-baseline", "--accuracy-steps", "0", "--no-exact",
           "--no-clip", "--no-side-blocks", "--no-pmc"]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "BM_FORCE_DISTRIBUTED"):
        env.pop(k, None)
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout)
        return read_counter_csvs(tmp, counter)
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def read_counter_csvs(root, counter):
    """{kernel name without its argument list: [dispatches, sum of the counter]} over every
    `*counter_collection.csv` rocprofv3 wrote under `root` (one row per dispatch and counter)."""
    import csv
    acc = {}
    for f in Path(root).rglob("*counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                a = acc.setdefault(k, [0, 0.0])
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return acc or None


def traffic_from_counters(fetch, write, kernel_label, child_steps):
    """(bytes per launch of the dominant kernel, its dispatches, HBM bytes per step over all kernels, top-12 table)
    from the two counter passes; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section)."""
    prefix = kernel_label.rstrip(">").replace(" ", "")
    per_launch = launches = None
    total = 0.0
    table = {}
    dom_bytes, dom_n = 0.0, 0
    for k, (n, f) in fetch.items():
        w = write.get(k)
        if w is None:
            continue
        b = (2.0 * f / n + w[1] / w[0]) * 1024.0
        total += b * n
        table[k] = {"launches_per_step": n / child_steps, "hbm_bytes_per_launch": b}
        if k.replace(" ", "").startswith(prefix):       # every instantiation of the label (e.g. <3,5,false> and <3,5,true>)
            dom_bytes += b * n
            dom_n += n
    if dom_n:
        per_launch, launches = dom_bytes / dom_n, dom_n
    top = dict(sorted(table.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_per_step"])[:12])
    return per_launch, launches, total / child_steps, top


def live_pmc_traffic(kernel_label, args, child_steps=5, timeout=90):
    """HBM traffic measured IN THIS RUN: two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one
    pass) over a short child of this very script, while this process keeps the GPU idle.  Corrections as the guide
    prescribes: both counters are in KiB and FETCH_SIZE reports half of the bytes of wide coalesced reads on gfx950
    -> bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Returns None when rocprofv3 is missing, this process is
    itself being profiled, or a pass fails -- the caller then falls back to the committed summary."""
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    t0 = time.perf_counter()
    fetch = _pmc_counter_csv("FETCH_SIZE", args, timeout)
    write = _pmc_counter_csv("WRITE_SIZE", args, timeout) if fetch else None
    if not fetch or not write:
        return None
    per_launch, launches, step_bytes, top = traffic_from_counters(fetch, write, kernel_label, child_steps)
    return {"hbm_bytes_per_launch": per_launch, "launches_counted": launches, "step_hbm_bytes": step_bytes,
            "child_steps": child_steps, "seconds": time.perf_counter() - t0, "per_kernel": top,
            "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over a child "
                   "`bench.py --steps 2 --warmup 1` of the same workload (1 warm-up + 2 timed + 2 event-pass steps = "
                   "5 steps; the child's set-up launches are counted into step_hbm_bytes: < 1 %); "
                   "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md"}


def cpu_baseline(seconds_budget=8.0, full_batch=True):
    """The CPU oracle (port of the reference path) on cfg1 = BASELINE.json configs[0] (fake-study shaped,
    batch 16), timed with 1 thread (bm/train.py:182) and with all host cores, bounded samples."""
    from oracle import bm_oracle as O
    c = synthetic.CONFIGS["cfg1"]
    B = c["B"]
    sb = synthetic.make_config_batch("cfg1", seed=2036)
    torch.manual_seed(0)
    model = SimpleConv(in_channels={"meg": c["C"]}, out_channels=c["F"], hidden={"meg": 320},
                       n_subjects=c["S"], **CLIP_CONV)
    sd = model.state_dict()
    ban = torch.tensor([0.5, 0.5])
    pos = sb.positions()
    prev = torch.get_num_threads()
    ncores = os.cpu_count() or prev
    # "all cores": torch's intra-op pool on every hardware thread of the GPU box's host loses to its own
    # synchronisation on this batch-16 job (256 threads: 83 s / step measured, slower than ONE thread), so
    # the multi-thread sample is capped at 32 threads; `host_cores` reports what the host has
    many = min(ncores, 32)
    out = {}
    for label, n in (("all_cores", many), ("one_thread", 1)):
        torch.set_num_threads(n)
        oracle = O.OracleModel({k: v.clone() for k, v in sd.items()}, O.CLIP_CONV_CFG, 320, c["F"])
        times, losses = [], []
        t_start = time.perf_counter()
        while len(times) < 2 or (time.perf_counter() - t_start < seconds_budget and len(times) < 7):
            t0 = time.perf_counter()
            losses.append(float(oracle.train_step(sb.meg, pos, sb.subject_index, sb.features, ban)[0]))
            times.append(time.perf_counter() - t0)
        timed = sorted(times[1:])
        med = timed[len(timed) // 2]
        out[label] = dict(value=B / med, unit="segments/s", cores=n, median_step_ms=med * 1e3,
                          steps_timed=len(timed), first_loss=losses[0])
    # the SAME configuration as `value` (cfg2, batch 256): 1 warm-up + 3 timed steps on `many` threads (~5 s each), the
    # median; and ONE thread (what the reference itself configures, bm/train.py:182) on a bounded sample of the same
    # workload: the first 32 segments of the batch (a whole batch-256 step takes ~20 s on one thread)
    cfg2 = None
    if full_batch:
        try:
            c2 = synthetic.CONFIGS["cfg2"]
            sb2 = synthetic.make_config_batch("cfg2", seed=2036)
            torch.manual_seed(0)
            m2 = SimpleConv(in_channels={"meg": c2["C"]}, out_channels=c2["F"], hidden={"meg": 320},
                            n_subjects=c2["S"], **CLIP_CONV)
            sd2 = m2.state_dict()
            torch.set_num_threads(many)
            o2 = O.OracleModel({k: v.clone() for k, v in sd2.items()}, O.CLIP_CONV_CFG, 320, c2["F"])
            pos2, t2 = sb2.positions(), []
            for _ in range(4):
                t0 = time.perf_counter()
                o2.train_step(sb2.meg, pos2, sb2.subject_index, sb2.features, ban)
                t2.append(time.perf_counter() - t0)
            med2 = sorted(t2[1:])[1]
            cfg2 = dict(value=c2["B"] / med2, unit="segments/s", cores=many, step_s=med2, steps_timed=3,
                        sample="cfg2 (C=208 T=360 F=120), batch 256 -- the configuration of `value` -- whole training "
                               "step of the CPU oracle, median of 3 steps after 1 warm-up")
            nb = 32
            torch.set_num_threads(1)
            o1 = O.OracleModel({k: v.clone() for k, v in sd2.items()}, O.CLIP_CONV_CFG, 320, c2["F"])
            t1 = []
            for _ in range(4):
                t0 = time.perf_counter()
                o1.train_step(sb2.meg[:nb], pos2[:nb], sb2.subject_index[:nb], sb2.features[:nb], ban)
                t1.append(time.perf_counter() - t0)
            med1 = sorted(t1[1:])[1]
            cfg2["one_thread"] = dict(value=nb / med1, unit="segments/s", cores=1, step_s=med1, steps_timed=3,
                                      sample=f"the first {nb} segments of the same cfg2 batch (bounded sample), "
                                             "median of 3 steps after 1 warm-up")
        except Exception as exc:            # the cfg1 figure must survive (host memory, time-outs)
            cfg2 = {"error": repr(exc)}
    torch.set_num_threads(prev)
    best = max(out.values(), key=lambda r: r["value"])
    cfg1 = dict(value=best["value"], unit="segments/s", cores=best["cores"],
                sample=f"cfg1 = BASELINE configs[0] (C=273 T=360 F=120, batch {B}), clip_conv model, whole training "
                       f"step, torch CPU fp32, median of {best['steps_timed']} steps after 1 warm-up",
                one_thread=out["one_thread"], all_cores=out["all_cores"])
    ref_file = ROOT / "profiles" / "r5_port_vs_reference.json"
    backing = None
    if ref_file.exists():
        try:
            backing = json.loads(ref_file.read_text())
            sub = backing.get("cfg2_b32") or {}
            backing = {k: backing[k] for k in ("port_over_reference", "reference_seg_per_s", "port_seg_per_s",
                                               "threads", "config", "where", "max_abs_loss_gap") if k in backing}
            backing["cfg2_shapes_batch32"] = {k: sub[k] for k in ("port_over_reference", "reference_seg_per_s",
                                                                  "port_seg_per_s", "max_abs_loss_gap") if k in sub}
            backing["file"] = "profiles/" + ref_file.name
        except Exception:
            backing = None
    note = ("CPU oracle = torch-CPU restatement of the reference path (kind \"port\": the reference modules need "
            "/root/reference, absent on the GPU box); `port_vs_reference` = the same step timed on the LIVE reference "
            "modules beside the port in the build container")
    if cfg2 and "value" in cfg2:
        # the configuration of `value` itself: like-for-like with the GPU number of this line
        return dict(value=cfg2["value"], unit="segments/s", cores=cfg2["cores"], kind="port", sample=cfg2["sample"],
                    step_s=cfg2["step_s"], steps_timed=cfg2["steps_timed"], one_thread=cfg2.get("one_thread"),
                    cfg1_b16=cfg1, host_cores=ncores, port_vs_reference=backing, note=note,
                    gpu_idle_and_no_profiler_children=True)
    return dict(value=cfg1["value"], unit="segments/s", cores=cfg1["cores"], kind="port",
                sample=cfg1["sample"] + " (the cfg2 batch-256 sample failed or was disabled: " + repr(cfg2) + ")",
                cfg1_b16=cfg1, cfg2_b256=cfg2, host_cores=ncores, port_vs_reference=backing, note=note)


def retrieval_block(workload, B, steps, dev, n_train=16, n_held=4, noise=3.4):
    """top-k segment retrieval (scripts/run_eval_probs.py:237-264 rule) of the full-size model after `steps`
    training steps on planted-latent synthetic batches, evaluated on held-out segments of the
    same synthetic world.  Chance level for top-10 is 10 / (n_held * B).  The planted noise keeps the
    task away from saturation."""
    from brainmagick_amd import retrieval
    from brainmagick_amd.losses import ClipLoss
    c = synthetic.CONFIGS[workload]
    kw = dict(planted=True, noise=noise)
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
            "train_steps": steps, "train_batches": n_train, "planted_noise": noise,
            "first_loss": losses[0], "last_loss": losses[-1],
            "data": "planted-latent synthetic world (brainmagick_amd/synthetic.py), default compute mode"}


# Dimensions of the long-horizon parity run: like tests/test_model_gpu.py::PARITY_DIMS, chosen so that every
# contraction runs in the kernels of the headline benchmark (wide f16x2 conv / weight gradient / score kernels).
PARITY_DIMS = dict(C=64, T=192, F=128, S=4, B=128, hidden=256, merger_channels=256, depth=4)
PARITY_NOISE = 2.5


def retrieval_parity_block(dev, steps=120, noise=PARITY_NOISE):
    """HIP path and CPU oracle trained side by side THROUGH THE HEADLINE KERNELS (same initial state, a fresh planted
    batch every step, model sized so that the wide f16x2 kernels cover its layers and the oracle still takes
    ~0.5 s / step): top-10 / top-1 retrieval on 1 024 held-out segments, largest relative loss gap."""
    import copy
    from oracle import bm_oracle as O
    from brainmagick_amd import retrieval
    from brainmagick_amd.losses import ClipLoss
    d = PARITY_DIMS
    cfg = dict(O.CLIP_CONV_CFG)
    cfg.update(merger_pos_dim=288, merger_channels=d["merger_channels"], initial_linear=d["merger_channels"],
               depth=d["depth"], merger_dropout=0.0)
    C, T, Fd, S, B, hidden, n_held = d["C"], d["T"], d["F"], d["S"], d["B"], d["hidden"], 1024
    torch.manual_seed(5)
    model = SimpleConv(in_channels={"meg": C}, out_channels=Fd, hidden={"meg": hidden}, n_subjects=S, **cfg)
    oracle = O.OracleModel(copy.deepcopy(model.state_dict()), cfg, hidden, Fd)
    solver = Solver(model, device=str(dev))
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(min(32, prev_threads))
    worst, kernels = 0.0, set()
    try:
        for step in range(steps):
            sb = synthetic.make_batch(B, C, T, Fd, S, seed=100 + step, planted=True, noise=noise)
            timer = None
            if step == steps - 1:
                timer = hip_ops.KernelTimer()
                hip_ops.set_kernel_timer(timer)
            lh = float(solver.train_step(sb))
            if timer is not None:
                hip_ops.set_kernel_timer(None)
                kernels = {r[0] for r in timer.records}
            lr_ = float(oracle.train_step(sb.meg, sb.positions(), sb.subject_index, sb.features)[0])
            worst = max(worst, abs(lh - lr_) / abs(lr_))
        held = synthetic.make_batch(n_held, C, T, Fd, S, seed=999, planted=True, noise=noise)
        est_hip, cand = solver.predict(held)
        acc_hip = retrieval.segment_topk_accuracy(ClipLoss().to(dev), est_hip, cand, topks=(1, 10))
        probs_ref = O.clip_probabilities(oracle.forward(held.meg, held.positions(), held.subject_index),
                                         held.features)
    finally:
        torch.set_num_threads(prev_threads)
        hip_ops.set_kernel_timer(None)
    labels = torch.arange(n_held)
    return {"hip_top10": acc_hip["top10"], "oracle_top10": O.topk_accuracy(probs_ref, labels, labels, 10),
            "hip_top1": acc_hip["top1"], "oracle_top1": O.topk_accuracy(probs_ref, labels, labels, 1),
            "max_rel_loss_gap": worst, "train_steps": steps, "held_out_segments": n_held,
            "planted_noise": noise, "kernels_of_the_last_step": sorted(kernels),
            "model": f"clip_conv reduced to the smallest size the wide f16x2 kernels cover (hidden {hidden}, depth "
                     f"{d['depth']}, C={C}, T={T}, F={Fd}), batch {B}, a fresh batch every step"}


def clip_roofline(dev, mode, reps=20):
    """The contrastive contraction scores = est . cand^T over K = F*T (bm/losses.py:91-95) at batch 256: the cfg2
    (F=120) and cfg3 (F=1024) shapes against the rank's own 256 candidates, and the cfg4 shapes against the 2 048
    whole-node candidates of 8 ranks (F=1024: a 3 GB operand, walked in row blocks).  HIP-event time of the whole
    forward (candidate norms NOT included: constant candidates; scores GEMM + split fold + row softmax / CE) and of
    the score kernel(s) alone."""
    from brainmagick_amd import functional as BF
    out = {}
    for name, F, Bc in (("cfg2", 120, 256), ("cfg3", 1024, 256), ("cfg4_mel_2048", 120, 2048),
                        ("cfg4_w2v_2048", 1024, 2048)):
        B, T = 256, 360
        K = F * T
        # DISTINCT operand sets, rotated, whose total exceeds the 256 MB memory-side cache: re-reading ONE 88 MB set
        # (cfg2) twenty times measures that cache, not HBM
        set_bytes = 4.0 * (B + Bc) * K
        nsets = max(1, min(4, int(-(-600e6 // set_bytes))))
        g = torch.Generator(device="cpu").manual_seed(3)
        sets = []
        for _ in range(nsets):
            est = torch.randn(B, F, T, generator=g).to(dev)
            cand = torch.empty(Bc, F, T, device=dev)
            for r0 in range(0, Bc, 256):
                cand[r0:r0 + 256] = torch.randn(256, F, T, generator=g).to(dev)
            if os.environ.get("BM_BENCH_ZERO_OPERANDS"):      # DVFS probe: same launches, operands that do not toggle
                est.zero_(), cand.zero_()
            sets.append((est, cand, hip_ops.clip_inv_norms(cand)))
        n_rep = reps if Bc == 256 else max(3, reps // 4)
        n_rep = -(-n_rep // nsets) * nsets
        for i in range(2 * nsets):
            BF.clip_forward_timed(*sets[i % nsets], None)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for i in range(n_rep):
            BF.clip_forward_timed(*sets[i % nsets], None)
        ev1.record()
        torch.cuda.synchronize()
        fwd_us = ev0.elapsed_time(ev1) / n_rep * 1e3
        timer = hip_ops.KernelTimer()
        for i in range(n_rep):
            BF.clip_forward_timed(*sets[i % nsets], timer)
        torch.cuda.synchronize()
        ks = timer.summary()
        t = sum(v["avg_ms"] * v["launches"] for v in ks.values()) / n_rep * 1e-3      # score kernels per forward
        kname = max(ks, key=lambda k: ks[k]["avg_ms"] * ks[k]["launches"])
        flops = 2.0 * B * Bc * K
        nbytes = 4.0 * (B + Bc) * K
        peak = MODE_PEAK_TFLOPS[mode]
        mfma_frac = flops / t / 1e12 / peak
        hbm_frac = nbytes / t / 1e9 / PEAK_HBM_GBS
        out[name] = {"kernel": kname, "launches_per_forward": sum(v["launches"] for v in ks.values()) / n_rep,
                     "B": B, "candidates": Bc, "K": K, "kernel_us": t * 1e6, "forward_us": fwd_us,
                     "achieved_tflops": flops / t / 1e12, "peak_tflops": peak, "mfma_frac": mfma_frac,
                     "achieved_gbs": nbytes / t / 1e9, "peak_gbs": PEAK_HBM_GBS, "hbm_frac": hbm_frac,
                     "bound": "mfma" if mfma_frac >= hbm_frac else "hbm", "frac": max(mfma_frac, hbm_frac),
                     "operand_sets_rotated": nsets, "operand_bytes_per_set": set_bytes}
        del est, cand, sets
        torch.cuda.empty_cache()
    return out


def retrieval_roofline(dev, mode, n_queries=2048, n_cand=10000, F=120, T=360, cpu_seconds=12.0):
    """Throughput of the batched retrieval evaluation (second half of the metric): `builds_probs`
    (scripts/run_eval_probs.py:267-307: probabilities of N' candidates for N predictions, in query blocks) +
    the top-10 / label-match rule (:237-264) at 2 048 queries x 10 000 candidates (bm/wer.py:71-79 draws 10 000
    negatives), HIP events around the whole evaluation.  Algorithmic bytes: the candidates once per query block, the
    queries once, the probabilities written and read once.  Beside it the reference's per-segment loop
    (bm/wer.py:91-116, restated in oracle.get_wer_loop) on a bounded sample of queries on the host cores."""
    from brainmagick_amd import retrieval
    from brainmagick_amd.losses import ClipLoss
    K = F * T
    g = torch.Generator(device="cpu").manual_seed(11)
    cand = torch.empty(n_cand, F, T, device=dev)
    for r0 in range(0, n_cand, 1000):
        cand[r0:r0 + 1000] = torch.randn(min(1000, n_cand - r0), F, T, generator=g).to(dev)
    est = torch.empty(n_queries, F, T, device=dev)
    for r0 in range(0, n_queries, 1024):
        est[r0:r0 + 1024] = torch.randn(min(1024, n_queries - r0), F, T, generator=g).to(dev)
    # plant the targets so that top-10 is neither 0 nor 1: query i resembles candidate i
    est += 0.015 * cand[:n_queries]
    clip = ClipLoss().to(dev)
    labels = torch.arange(n_cand)
    block = 1024

    def run():
        probs = retrieval.builds_probs(clip, est, cand, batch_size=block)
        return retrieval.get_accuracy_from_probs(probs, labels[:n_queries], labels, topk=10)

    run()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    ev0.record()
    for _ in range(reps):
        top10 = run()
    ev1.record()
    torch.cuda.synchronize()
    t = ev0.elapsed_time(ev1) / reps * 1e-3
    nblocks = -(-n_queries // block)
    nbytes = 4.0 * (nblocks * n_cand * K + n_queries * K + 2.0 * n_queries * n_cand)
    flops = 2.0 * n_queries * n_cand * K
    peak = MODE_PEAK_TFLOPS[mode]
    out = {"queries": n_queries, "candidates": n_cand, "K": K, "query_block": block, "seconds": t,
           "queries_per_s": n_queries / t, "top10": top10,
           "achieved_gbs": nbytes / t / 1e9, "peak_gbs": PEAK_HBM_GBS, "hbm_frac": nbytes / t / 1e9 / PEAK_HBM_GBS,
           "achieved_tflops": flops / t / 1e12, "peak_tflops": peak, "mfma_frac": flops / t / 1e12 / peak,
           "algorithmic_bytes": nbytes}
    out["bound"] = "mfma" if out["mfma_frac"] >= out["hbm_frac"] else "hbm"
    out["frac"] = max(out["mfma_frac"], out["hbm_frac"])
    # the reference's loop (one softmax over all candidates per test segment) on the host, bounded sample
    try:
        from oracle import bm_oracle as O
        cand_h, est_h = cand.cpu(), est[:64].cpu()
        hashes = torch.arange(n_cand)
        kept = torch.arange(n_cand)
        prev = torch.get_num_threads()
        torch.set_num_threads(min(32, os.cpu_count() or prev))
        t0 = time.perf_counter()
        O.get_wer_loop(est_h[:2], cand_h, hashes, kept)           # calibration (includes the one-off clone)
        per_query = (time.perf_counter() - t0) / 2
        done = max(4, min(len(est_h), int(cpu_seconds / max(per_query, 1e-3))))
        t0 = time.perf_counter()
        O.get_wer_loop(est_h[:done], cand_h, hashes, kept)
        dt = time.perf_counter() - t0
        torch.set_num_threads(prev)
        out["cpu_loop"] = {"queries_per_s": done / dt, "queries_timed": done, "cores": min(32, os.cpu_count() or prev),
                           "kind": "port", "sample": f"oracle.get_wer_loop (bm/wer.py:91-116 as written) on {done} "
                                                     f"queries against the same {n_cand} candidates"}
    except Exception as exc:
        out["cpu_loop"] = {"error": repr(exc)}
    del est, cand
    torch.cuda.empty_cache()
    return out


def pcie_inclusive_block(workload, B, dev, solver, steps, rank=0, n=4):
    """The same step with the batches handed over as PINNED HOST tensors (what a DataLoader with pin_memory=True
    delivers; bm/solver.py:243 `batch.to(device)`): Solver.stage copies batch k + 1 on the copy stream while step k
    runs.  Never `value`."""
    c = synthetic.CONFIGS[workload]
    host = [synthetic.make_batch(B, c["C"], c["T"], c["F"], c["S"], seed=7000 + 1000 * rank + i,
                                 mixed_eeg=workload == "cfg5").pin() for i in range(n)]
    mb = sum(t.numel() * t.element_size() for t in (host[0].meg, host[0].features, host[0].features_mask)) / 1e6
    for i in range(3):
        solver.train_step(host[i % n], next_batch=host[(i + 1) % n])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = solver.train_step(host[(i + 3) % n], next_batch=host[(i + 4) % n])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    if not bool(torch.isfinite(loss)):
        raise RuntimeError("non-finite loss in the pcie-inclusive block")
    return {"ms_per_step": ms, "value": B * 1e3 / ms, "unit": "segments/s", "steps": steps,
            "host_to_device_MB_per_step": mb, "final_loss": float(loss),
            "how": "pinned host batches, next batch staged on a copy stream during the step (Solver.stage / prefetch)"}


def flush_c_stdio():
    """librccl prints a version banner to the C-level stdout when the first communicator comes up; on a pipe that
    text sits in the stdio buffer until exit and would land BEHIND the JSON line.  Flushing right after init (every
    rank) and right before the result keeps the JSON line the last thing on stdout."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def self_launch_command(gpus: int, argv, port=None):
    """(command, environment) of the re-launch under torch.distributed.run: one rank per GPU of this node,
    rendezvous on 127.0.0.1 (the container hostname may not resolve) on a free port."""
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    return cmd, env


def self_launch(args) -> int:
    """python bench.py --gpus N without a torchrun environment: re-launch under torch.distributed.run."""
    cmd, env = self_launch_command(args.gpus, sys.argv[1:])
    return subprocess.call(cmd, env=env)


class BatchStream:
    """`n` distinct batches resident in HBM; every step gets the next one with a FRESH draw of the
    segment -> recording assignment (host side), like a shuffled training stream would produce."""

    def __init__(self, workload, B, rank, dev, n=8, n_recordings=16, n_draws=64):
        c = synthetic.CONFIGS[workload]
        gen = torch.Generator().manual_seed(4242)
        if workload == "cfg5":
            layouts = [torch.rand(c["C"], 2, generator=gen), torch.rand(128, 2, generator=gen)]
            self.pool = [synthetic.Recording(i, layouts[i % 2].clone()) for i in range(n_recordings)]
        else:       # one sensor layout shared by every recording of the study (gwilliams2022: one MEG system)
            base = torch.rand(c["C"], 2, generator=gen)
            self.pool = [synthetic.Recording(i, base.clone()) for i in range(n_recordings)]
        self.batches = []
        for i in range(n):
            sb = synthetic.make_batch(B, c["C"], c["T"], c["F"], c["S"], seed=2036 + 1000 * rank + i,
                                      recordings=self.pool, mixed_eeg=workload == "cfg5")
            self.batches.append(sb.to(dev))
        # fresh assignments that keep each segment on a recording of its own layout class
        g2 = torch.Generator().manual_seed(99 + rank)
        self.draws = []
        for d in range(n_draws):
            sb = self.batches[d % n]
            recs = []
            for r in sb._recordings:
                same = [q for q in self.pool if len(q.layout) == len(r.layout)]
                recs.append(same[int(torch.randint(0, len(same), (1,), generator=g2))])
            self.draws.append(recs)
        self.n = n
        self.step = 0
        self._peeked = None

    def _at(self, step):
        sb = self.batches[step % self.n]
        return sb.replace(_recordings=self.draws[step % len(self.draws)])

    def next(self):
        """(this step's batch, the next step's batch): the second one lets the Solver start the whole-node
        candidate all-gather of step k + 1 behind step k's backward (Solver.prefetch)."""
        if self._peeked is None:
            self._peeked = self._at(self.step)
        cur = self._peeked
        self.step += 1
        self._peeked = self._at(self.step)
        return cur, self._peeked


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
                    help="skip the extra timing blocks of the other fp32-class modes (profiling runs)")
    ap.add_argument("--no-clip", action="store_true", help="skip the ClipLoss contraction roofline block")
    ap.add_argument("--no-side-blocks", action="store_true",
                    help="skip the sustained / single-stream / pcie-inclusive / retrieval-roofline blocks (A/B and "
                         "profiling runs)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not spawn the rocprofv3 --pmc children that measure `roofline.traffic` in this run")
    ap.add_argument("--sustained-steps", type=int, default=200)
    ap.add_argument("--dtype", default=hip_ops.DEFAULT_COMPUTE_DTYPE, choices=sorted(hip_ops.COMPUTE_DTYPES),
                    help="compute mode of the contractions; every fp32-class mode (f16x2, f32x3, f32) is held to the "
                         "same parity tolerances")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))

    hip_ops.set_compute_dtype(args.dtype)
    distrib.init()
    flush_c_stdio()
    rank, world = distrib.rank(), distrib.world_size()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    stream = BatchStream(args.workload, B, rank, dev)
    zero_operands = bool(os.environ.get("BM_BENCH_ZERO_OPERANDS"))
    if zero_operands:
        # DVFS probe (never a result): the same launches, addresses and memory traffic on operands that do not toggle
        # -- all parameters and inputs zero -- show what the power manager takes from the matrix kernels on real data
        with torch.no_grad():
            for p_ in model.parameters():
                p_.zero_()
        hip_ops.weights_changed()
        for sb in stream.batches:
            sb.meg.zero_()
            sb.features.zero_()

    def timed_steps(k):
        # (a full Python garbage collection costs this process ~70 ms once, 30-40 steps in: keep it out of whichever
        # timed pass it would fall into -- it is not part of a step)
        gc.collect()
        distrib.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            cur, nxt = stream.next()
            loss = solver.train_step(cur, next_batch=nxt if prefetch else None)
        distrib.barrier()
        torch.cuda.synchronize()
        return distrib.max_over_ranks(time.perf_counter() - t0), loss

    prefetch = negatives == "node" and distrib.is_distributed()       # (also under BM_FORCE_DISTRIBUTED=1 at world 1)
    for _ in range(args.warmup):
        cur, nxt = stream.next()
        solver.train_step(cur, next_batch=nxt if prefetch else None)
    elapsed, loss = timed_steps(args.steps)
    final_loss = float(loss)
    if final_loss != final_loss or abs(final_loss) == float("inf"):
        # a throughput number of a diverged run is worthless (and flattering: the matrix pipe clocks higher on NaNs)
        raise SystemExit(f"bench: the training loss is not finite after the timed region ({final_loss})")

    # second pass over the same stream with per-launch HIP events (NOT part of `value`)
    timer = hip_ops.KernelTimer()
    hip_ops.set_kernel_timer(timer)
    comm_timer = distrib.CommTimer() if distrib.is_distributed() else None
    distrib.set_comm_timer(comm_timer)
    event_pass_elapsed, _ = timed_steps(args.steps)
    hip_ops.set_kernel_timer(None)
    distrib.set_comm_timer(None)
    comm_ms = None
    if comm_timer is not None:
        torch.cuda.synchronize()
        comm_ms = comm_timer.summary(args.steps)
    sustained = pcie = None
    if not args.no_side_blocks:
        # `value` comes from the driver's fixed --steps (0.3 s at 20 steps): the same loop over a longer window
        solver.train_step(stream.next()[0])
        e_s, _ = timed_steps(args.sustained_steps)
        sustained = {"steps": args.sustained_steps, "ms_per_step": e_s / args.sustained_steps * 1e3,
                     "value": world * B * args.sustained_steps / e_s, "unit": "segments/s",
                     "seconds": e_s}
        if world == 1:
            try:
                pcie = pcie_inclusive_block(args.workload, B, dev, solver, args.steps, rank)
                pcie["resident_ms_per_step"] = elapsed / args.steps * 1e3
            except Exception as exc:
                pcie = {"error": repr(exc)}

    # For transparency the same job is also timed in the other fp32-class modes (not part of `value`).
    other_modes = {}
    if not args.no_exact and args.dtype in ("f16x2", "f32x3"):
        for mode in ("f16x2", "f32x3", "f32"):
            if mode == args.dtype or mode not in hip_ops.COMPUTE_DTYPES:
                continue
            hip_ops.set_compute_dtype(mode)
            k2 = max(2, args.steps // 4)
            solver.train_step(stream.next()[0])
            e2, _ = timed_steps(k2)
            other_modes[mode] = {"compute_mode": mode, "steps": k2, "ms_per_step": e2 / k2 * 1e3,
                                 "value": world * B * k2 / e2, "unit": "segments/s"}
        hip_ops.set_compute_dtype(args.dtype)

    # DVFS probe (last use of this solver; never a result): the same launches, addresses and memory traffic on operands
    # that do not toggle -- every parameter, Adam moment and input zero.  What the step gains is what the power manager
    # takes from the matrix kernels on real data.
    dvfs = None
    if not args.no_side_blocks and not zero_operands:
        with torch.no_grad():
            for p_ in model.parameters():
                p_.zero_()
            solver.optimizer.exp_avg.zero_()
            solver.optimizer.exp_avg_sq.zero_()
        hip_ops.weights_changed()
        for sb in stream.batches:
            sb.meg.zero_()
            sb.features.zero_()
        for _ in range(3):
            cur, nxt = stream.next()
            solver.train_step(cur, next_batch=nxt if prefetch else None)
        e_z, loss_z = timed_steps(args.steps)
        dvfs = {"zero_operand_ms_per_step": e_z / args.steps * 1e3, "ms_per_step": elapsed / args.steps * 1e3,
                "ratio": elapsed / e_z, "zero_operand_loss": float(loss_z),
                "what": "the timed loop repeated with all parameters, Adam moments and inputs zero: identical launches "
                        "and memory traffic, operands that do not toggle; the gain is the clock the power manager "
                        "takes away on real data"}

    if rank != 0:
        distrib.barrier()
        return
    ksum = timer.summary()
    dom = max(ksum, key=lambda k: ksum[k]["avg_ms"] * ksum[k]["launches"])
    d = ksum[dom]
    achieved = d["flops_per_launch"] / (d["avg_ms"] * 1e-3) / 1e12
    seg_flops = algorithmic_flops_per_segment(c["C"], c["T"], c["F"])
    traffic, traffic_src, traffic_avg_ms = pmc_traffic(dom)
    # HBM traffic measured in THIS run (rocprofv3 children while this process leaves the GPU alone); they overlap with the
    # CPU baseline, which keeps the GPU idle anyway.  Collected further down; the committed summary is the fallback.
    pmc_box = {}
    pmc_thread = None
    if world == 1 and not args.no_pmc and not args.no_side_blocks and not zero_operands:
        import threading
        torch.cuda.synchronize()
        pmc_thread = threading.Thread(target=lambda: pmc_box.update(live=live_pmc_traffic(dom, args)), daemon=True)
    # counters are collected in separate rocprofv3 runs and committed: a summary whose launch duration for this
    # kernel is off by more than 10 % from what has just been measured describes an older kernel
    traffic_stale = bool(traffic is not None and traffic_avg_ms and
                         abs(traffic_avg_ms - d["avg_ms"]) > 0.10 * d["avg_ms"])
    peak_tf = MODE_PEAK_TFLOPS[args.dtype]
    total_kernel_ms = sum(v["avg_ms"] * v["launches"] for v in ksum.values()) / args.steps
    step_bytes = algorithmic_bytes_per_step(c["C"], c["T"], c["F"], B, n_params)
    out = {
        "metric": ("segments/s, 208-ch x 360-sample SimpleConv + ClipLoss training step" if not zero_operands else
                   "DVFS PROBE ON ALL-ZERO OPERANDS -- not a result"),
        "value": world * B * args.steps / elapsed,
        "unit": "segments/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": MODE_DTYPE[args.dtype],
        "compute_mode": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.workload}: gwilliams2022-shaped MEG C={c['C']} T={c['T']}, "
                               f"F={c['F']} features, {c['S']} subjects, clip_conv SimpleConv "
                               f"({n_params} params) + ClipLoss + Adam",
                   "batch_per_gpu": B, "global_batch": world * B, "negatives": negatives,
                   "parallelism": f"dp{world}", "comm": distrib.comm_kind(), "rccl_world": distrib.reported_world(),
                   "candidate_gather": ("prefetched one step ahead on a side stream" if prefetch else
                                        ("in front of the forward" if negatives == "node" else "none")),
                   "distinct_batches": stream.n, "final_loss": final_loss,
                   "step_tflops": seg_flops * B * world * args.steps / elapsed / 1e12},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved,
                     "peak": peak_tf, "unit": "TFLOP/s", "frac": achieved / peak_tf,
                     # information: the same ratio against what a register-only MFMA loop on random 16-bit operands
                     # sustains on this chip (1 790 TF at a power-throttled ~1.75 GHz, scripts/micro/mfma_peak.hip)
                     "frac_of_sustained_mfma": (achieved / (peak_tf * 1790.0 / 2500.0)
                                                if args.dtype in ("f16x2", "f32x3") else None),
                     "traffic": traffic,
                     "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, separate passes)",
                     "traffic_source": "committed:" + str(traffic_src), "traffic_stale": traffic_stale,
                     "traffic_summary_avg_launch_ms": traffic_avg_ms,

                     "avg_launch_ms": d["avg_ms"], "launches_per_step": d["launches"] / args.steps,
                     "median_launch_ms": d["median_ms"],
                     "event_outliers_dropped": {k: v["outliers"] for k, v in ksum.items() if v["outliers"]},
                     "algorithmic_flops_per_launch": d["flops_per_launch"],
                     "mfma_kernels_ms_per_step": total_kernel_ms,
                     "event_pass_ms_per_step": event_pass_elapsed / args.steps * 1e3,
                     "per_kernel_ms_per_step": {k: v["avg_ms"] * v["launches"] / args.steps for k, v in
                                                sorted(ksum.items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches"])},
                     "hbm_roofline_frac_step": step_bytes / (PEAK_HBM_GBS * 1e9) / (elapsed / args.steps),
                     "algorithmic_bytes_per_step": step_bytes},
    }
    # per-phase communication times of the event pass (HIP events on the stream each collective is enqueued on): what a
    # first real multi-GPU run needs to explain itself.  `cand_gather` runs on the side stream under the backward pass;
    # `gather_wait` is the part of it the compute stream actually waits for.
    out["comm_ms"] = comm_ms
    out["comm_world_reported"] = distrib.reported_world()
    # one-pass BatchNorm backward: workgroups that gave up polling for a partner's sums and computed them themselves
    # (slow path, never wrong) since the process started -- 0 unless partners were not co-resident
    out["bn_bwd_fused_fallback_workgroups"] = int(_lib.lib().bm_act_bn_bwd_fused_fallbacks())
    out["dvfs_probe"] = dvfs
    out["sustained"] = sustained
    out["pcie_inclusive"] = pcie
    out["other_fp32_modes"] = other_modes or None
    out["exact_f32_mfma"] = other_modes.get("f32")
    out["roofline_clip"] = None
    if not args.no_clip:
        try:
            out["roofline_clip"] = clip_roofline(dev, args.dtype)
        except Exception as exc:
            out["roofline_clip"] = {"error": repr(exc)}
    out["roofline_retrieval"] = None
    if world == 1 and not args.no_side_blocks and args.workload == "cfg2":
        try:
            out["roofline_retrieval"] = retrieval_roofline(dev, args.dtype)
        except Exception as exc:
            out["roofline_retrieval"] = {"error": repr(exc)}
    out["retrieval"] = None
    out["retrieval_parity"] = None
    if world == 1 and args.accuracy_steps > 0:
        try:
            out["retrieval"] = retrieval_block(args.workload, B, args.accuracy_steps, dev)
        except Exception as exc:   # the throughput line must survive a failure of the side measurement
            out["retrieval"] = {"error": repr(exc)}
        try:
            out["retrieval_parity"] = retrieval_parity_block(dev)
        except Exception as exc:
            out["retrieval_parity"] = {"error": repr(exc)}
    if pmc_thread is not None:
        # the counter children run ALONE: nothing of this process on the GPU, and the CPU baseline starts only when they
        # are done (they are Python processes of their own: next to it they took host cores from the number it reports)
        torch.cuda.synchronize()
        pmc_thread.start()
        pmc_thread.join(timeout=200)      # (measured: 9.5 s for both passes; a pass that hangs is killed after 90 s)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    else:
        out["cpu_baseline"] = None
    if pmc_thread is not None:
        live = pmc_box.get("live")
        r = out["roofline"]
        if live and live.get("hbm_bytes_per_launch"):
            r["traffic_committed"] = {"traffic": r["traffic"], "source": r["traffic_source"], "stale": r["traffic_stale"]}
            r["traffic"] = live["hbm_bytes_per_launch"]
            r["traffic_source"] = "live"
            r["traffic_stale"] = False
            r["traffic_live"] = live
            r["step_hbm_bytes"] = live["step_hbm_bytes"]
    distrib.barrier()
    flush_c_stdio()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    try:
        main()
    finally:
        distrib.shutdown()
