import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import brainmagick_amd
from brainmagick_amd import hip_ops as H
brainmagick_amd.set_compute_dtype("f32x3")
B, T = 256, 360
for (M, Cin, dil) in [(320, 320, 4), (320, 320, 8), (640, 320, 1), (320, 640, 1)]:
    x = torch.randn(B, Cin, T, device="cuda")
    dy = torch.randn(B, M, T, device="cuda")
    print("launch", M, Cin, dil, flush=True)
    out = H.gemm_nt(dy, x, B, M, Cin, T, 3, dil)
    torch.cuda.synchronize()
    ref = torch.zeros(M, Cin, 3, dtype=torch.float64, device="cuda")
    xd, dyd = x.double(), dy.double()
    for j in range(3):
        s = (j - 1) * dil
        xs = torch.zeros_like(xd)
        if s >= 0:
            xs[:, :, :T - s] = xd[:, :, s:]
        else:
            xs[:, :, -s:] = xd[:, :, :T + s]
        ref[:, :, j] = torch.einsum("bmt,bct->mc", dyd, xs)
    print("  rel", ((out[0].double() - ref).norm() / ref.norm()).item(), flush=True)
