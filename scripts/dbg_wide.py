import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import brainmagick_amd
from brainmagick_amd import hip_ops as H
brainmagick_amd.set_compute_dtype("f32x3")
torch.manual_seed(0)
cases = [tuple(int(v) for v in c.split(',')) for c in sys.argv[1:]]
for (B, Cin, M, KS, dil, T) in cases:
    x = torch.randn(B, Cin, T, device="cuda")
    w = torch.randn(M, Cin, KS, device="cuda") / (Cin * KS) ** 0.5
    b = torch.randn(M, device="cuda")
    wp = H.pack_conv_fwd(w)
    print("launch", B, Cin, M, KS, dil, T, flush=True)
    y = H.conv_nn(x, wp, M, KS, dil, bias=b, want_pre=True, want_out=False)
    y = y[0] if isinstance(y, (tuple, list)) else y
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv1d(x.double(), w.double(), b.double(), padding=(KS // 2) * dil, dilation=dil)
    err = ((y.double() - ref).norm() / ref.norm()).item()
    print("rel_l2", err, flush=True)
