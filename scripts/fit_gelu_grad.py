"""Coefficients and fp32 error budget of ``bm_gelu_grad_fast`` (csrc/bm_common.h).

    Phi(-a) = exp(-a^2 / 2) * Q(t),  t = 1 / (1 + p a),  Q = sum_{k=1..7} c_k t^k      (a >= 0)

fitted by iteratively re-weighted least squares to the ABSOLUTE error of e * Q (what enters gelu'), then the
whole derivative is emulated element by element in fp32 (every operation rounded to fp32, fused multiply-adds
as fused) against fp64 and beside torch's own fp32 GELU backward -- the reference's arithmetic.
Run on the CPU: ``python scripts/fit_gelu_grad.py``."""
import numpy as np
import torch
from scipy import special

P, DEG = 0.24, 7


def fit(p=P, deg=DEG, amax=14.0, n=40001):
    a = np.linspace(0, amax, n)
    t = 1 / (1 + p * a)
    e = np.exp(-a * a / 2)
    target = special.ndtr(-a)
    A = np.stack([e * t ** k for k in range(1, deg + 1)], 1)
    w = np.ones(n)
    for _ in range(60):
        c, *_ = np.linalg.lstsq(A * w[:, None], target * w, rcond=None)
        err = A @ c - target
        w = w * (1 + 4 * np.abs(err) / np.abs(err).max())
        w /= w.mean()
    return c, np.abs(A @ c - target).max()


f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def gelu_grad_fast(z, c):
    z = z.astype(f32)
    a = np.abs(z)
    t = (f32(1) / fma(a, np.full_like(a, f32(P)), np.ones_like(a))).astype(f32)
    e = np.exp2(((z * z).astype(f32) * f32(-0.5 * 1.4426950408889634)).astype(np.float64)).astype(f32)
    q = np.full_like(a, f32(c[-1]))
    for k in range(len(c) - 2, -1, -1):
        q = fma(q, t, np.full_like(a, f32(c[k])))
    eq = (e * (q * t).astype(f32)).astype(f32)
    base = np.where(z >= 0, f32(1) - eq, eq).astype(f32)
    return fma(e, (z * f32(0.3989422804014327)).astype(f32), base)


if __name__ == "__main__":
    c, m = fit()
    print("p =", P, "coefficients c1..c7 =", [float(v) for v in c], "max |e Q - Phi(-a)| =", m)
    z = np.concatenate([np.linspace(-12, 12, 2000001), np.random.default_rng(0).standard_normal(1000000) * 2]).astype(f32)
    zd = z.astype(np.float64)
    ref = special.ndtr(zd) + zd * np.exp(-zd * zd / 2) / np.sqrt(2 * np.pi)
    g = gelu_grad_fast(z, c)
    zt = torch.tensor(z, requires_grad=True)
    torch.nn.functional.gelu(zt).sum().backward()
    gt = zt.grad.numpy()
    for name, v in (("bm_gelu_grad_fast (fp32 emulation)", g), ("torch fp32 GELU backward", gt)):
        print(f"{name:38s} max abs err {np.abs(v - ref).max():.3e}   rms {np.sqrt(np.mean((v - ref) ** 2)):.3e}")
