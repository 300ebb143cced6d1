"""Error budget of Winograd F(2,3) along time for the dilated 3-tap convs of the stack (CPU, torch): two outputs
(t, t + d) from the inputs (t - d, t, t + d, t + 2d) with 4 products instead of 6.  Prints the relative L2 error of the
fp32 Winograd evaluation and of torch's direct fp32 conv against an fp64 reference (DESIGN.md section 5, round 4:
"the lever that is left").  Usage: python scripts/winograd_f23_error.py"""
import torch
import torch.nn.functional as F


def winograd_f23(x, w, d):
    """x [B, C, T] (T a multiple of 2 d), w [M, C, 3], 'same' zero padding, dilation d."""
    B, C, T = x.shape
    xp = F.pad(x, (d, 2 * d))                       # index t + d <-> input t; room for t + 2 d
    # tiles: base positions t with (t mod 2 d) < d; outputs at t and t + d
    t = torch.arange(T)
    base = t[(t % (2 * d)) < d]
    d0, d1, d2, d3 = (xp[:, :, base + k * d] for k in range(4))          # inputs t - d, t, t + d, t + 2 d
    g0, g1, g2 = w[:, :, 0], w[:, :, 1], w[:, :, 2]
    u = (g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2)             # G g, once per step
    v = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)                             # B^T d, fp32 adds
    m = [torch.einsum("mc,bcn->bmn", u[k], v[k]) for k in range(4)]
    y = torch.empty(B, w.shape[0], T, dtype=x.dtype)
    y[:, :, base] = m[0] + m[1] + m[2]
    y[:, :, base + d] = m[1] - m[2] - m[3]
    return y


def main():
    torch.manual_seed(0)
    B, C, M, T = 4, 320, 320, 384
    for d in (1, 2, 4, 8, 16):
        x = torch.randn(B, C, T, dtype=torch.float64)
        w = torch.randn(M, C, 3, dtype=torch.float64) / (3 * C) ** 0.5
        ref = F.conv1d(x, w, padding=d, dilation=d)
        chk = winograd_f23(x, w, d)
        assert (chk - ref).norm() / ref.norm() < 1e-12, "the restatement itself"
        wino = winograd_f23(x.float(), w.float(), d).double()
        direct = F.conv1d(x.float(), w.float(), padding=d, dilation=d).double()
        rel = lambda a: ((a - ref).norm() / ref.norm()).item()
        print(f"dilation {d:2d}: Winograd F(2,3) fp32 {rel(wino):.2e}   direct fp32 {rel(direct):.2e}")


if __name__ == "__main__":
    main()
