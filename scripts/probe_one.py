import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import brainmagick_amd
from brainmagick_amd import hip_ops as H
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
brainmagick_amd.set_compute_dtype(mode)
B, T, Cin, M, KS, dil = 256, 360, 320, 320, 3, 1
x = torch.randn(B, Cin, T, device="cuda")
w = torch.randn(M, Cin, KS, device="cuda") / (Cin * KS) ** 0.5
b = torch.randn(M, device="cuda")
dy = torch.randn(B, M, T, device="cuda")
wp = H.pack_conv_fwd(w)
for _ in range(5):
    H.conv_nn(x, wp, M, KS, dil, bias=b, want_pre=True, want_out=False)
    H.gemm_nt(dy, x, B, M, Cin, T, KS, dil)
torch.cuda.synchronize()
