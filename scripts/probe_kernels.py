"""Kernel-level timing probe (HIP events, interleaved modes): the conv / weight-gradient shapes of the
clip_conv model at B=256, T=360.  usage: python scripts/probe_kernels.py [conv|wgrad|clip] [modes...]"""
import sys
import math
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import brainmagick_amd  # noqa: E402
from brainmagick_amd import hip_ops as H  # noqa: E402

import os
WANT_PRE = os.environ.get("PROBE_OUT", "0") != "1"
what = sys.argv[1] if len(sys.argv) > 1 else "conv"
modes = sys.argv[2:] or ["f32x3", "f16x2"]
B, T = 256, 360
dev = "cuda"
g = torch.Generator().manual_seed(0)


def timeit(fn, reps=12):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e-3


if what == "conv":
    shapes = [(320, 320, 3, 2), (320, 640, 3, 1), (270, 320, 3, 1), (320, 640, 1, 1), (640, 120, 1, 1),
              (208, 270, 1, 1), (270, 270, 1, 1), (270, 208, 1, 1)]
    for Cin, M, KS, dil in shapes:
        x = torch.randn(B, Cin, T, generator=g).to(dev)
        w = (torch.randn(M, Cin, KS, generator=g) / math.sqrt(Cin * KS)).to(dev)
        res = {}
        for rnd in range(2):
            for mode in modes:
                brainmagick_amd.set_compute_dtype(mode)
                wp = H.pack_conv_fwd(w, (T, dil))
                H.amax(x)
                t = timeit(lambda: H.conv_nn(x, wp, M, KS, dil, want_pre=WANT_PRE, want_out=not WANT_PRE))
                res.setdefault(mode, []).append(t)
        fl = 2.0 * B * T * M * Cin * KS
        print(f"conv Cin={Cin} M={M} KS={KS} d={dil}: " + "  ".join(
            f"{m}: {min(v) * 1e6:7.1f} us {fl / min(v) / 1e12:6.1f} TF" for m, v in res.items()), flush=True)
elif what == "wgrad":
    shapes = [(320, 320, 3, 2), (320, 640, 3, 1), (270, 320, 3, 1), (320, 640, 1, 1), (270, 270, 1, 1)]
    for Cin, M, KS, dil in shapes:
        x = torch.randn(B, Cin, T, generator=g).to(dev)
        dy = torch.randn(B, M, T, generator=g).to(dev)
        res = {}
        for rnd in range(2):
            for mode in modes:
                brainmagick_amd.set_compute_dtype(mode)
                H.amax(x), H.amax(dy)
                t = timeit(lambda: H.gemm_nt(dy, x, B, M, Cin, T, KS, dil))
                res.setdefault(mode, []).append(t)
        fl = 2.0 * B * T * M * Cin * KS
        print(f"wgrad Cin={Cin} M={M} KS={KS} d={dil}: " + "  ".join(
            f"{m}: {min(v) * 1e6:7.1f} us {fl / min(v) / 1e12:6.1f} TF" for m, v in res.items()), flush=True)
elif what == "clip":
    from brainmagick_amd import functional as BF
    for F_ in (120, 1024):
        est = torch.randn(B, F_, T, generator=g).to(dev).requires_grad_()
        cand = torch.randn(B, F_, T, generator=g).to(dev)
        for mode in modes:
            brainmagick_amd.set_compute_dtype(mode)
            inv = H.clip_inv_norms(cand)
            t_f = timeit(lambda: BF.clip_forward_timed(est.detach(), cand, inv, None))

            def fb():
                loss = BF.ClipLossFn.apply(est, cand, 0)[0]
                loss.backward()
            t_fb = timeit(fb, reps=6)
            print(f"clip F={F_} {mode}: forward {t_f * 1e6:.1f} us, fwd+bwd (incl. norms, autograd) {t_fb * 1e6:.1f} us",
                  flush=True)
