"""Is the wide f16x2 conv bound by its schedule or by the chip's power budget?  Runs the production kernel on random
operands and on all-zero operands: identical instruction stream, addresses and memory traffic; only the toggling in
the matrix cores differs (MI355X_MICROARCH.md, "DVFS give-back").  Results of round 3: profiles/r3_ab_notes.md.

    python scripts/probe_conv_dvfs.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brainmagick_amd import hip_ops as H  # noqa: E402

SHAPES = [  # Cin, M, KS, dil, residual
    (320, 320, 3, 1, True), (320, 320, 3, 2, True), (320, 320, 3, 16, False), (320, 640, 3, 1, False),
    (640, 320, 3, 1, False), (270, 320, 3, 1, False), (320, 640, 1, 1, False), (270, 270, 1, 1, False)]
B, T = 256, 360


def time_conv(x, w, bias, res, M, KS, dil, reps=10):
    wp = H.pack_weights(w, 1, M, w.shape[1], KS, 0, w.shape[1] * KS, KS, 1, shape=(T, dil))
    H.amax(x)
    for _ in range(3):
        H.conv_nn(x, wp, M, KS, dil, bias=bias, res=res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        H.conv_nn(x, wp, M, KS, dil, bias=bias, res=res)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    H.set_compute_dtype("f16x2")
    print(f"{'Cin, M, KS, dil, residual':30s} {'random us':>10s} {'zeros us':>10s} {'change':>8s}")
    for Cin, M, KS, dil, with_res in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(Cin * 7 + M + KS + dil)
        x = torch.randn(B, Cin, T, device="cuda", generator=g)
        w = torch.randn(M, Cin, KS, device="cuda", generator=g) / (Cin * KS) ** 0.5
        bias = torch.randn(M, device="cuda", generator=g)
        res = torch.randn(B, M, T, device="cuda", generator=g) if with_res else None
        t_random = time_conv(x, w, bias, res, M, KS, dil)
        zx, zw, zb = torch.zeros_like(x), torch.zeros_like(w), torch.zeros_like(bias)
        t_zero = time_conv(zx, zw, zb, None if res is None else torch.zeros_like(res), M, KS, dil)
        print(f"{str((Cin, M, KS, dil, with_res)):30s} {t_random:10.1f} {t_zero:10.1f} {100 * (t_zero / t_random - 1):+7.1f}%")


if __name__ == "__main__":
    main()
