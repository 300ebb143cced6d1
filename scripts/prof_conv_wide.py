"""Per-stage cycle breakdown of conv_nn_x3w_kernel (library built with -DWIDE_PROFILE; BM_HIP_LIB=...)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import brainmagick_amd
from brainmagick_amd import hip_ops as H
brainmagick_amd.set_compute_dtype("f32x3")
B, T = 256, 360
for (Cin, M, KS, dil) in [(320, 320, 3, 1), (320, 320, 3, 16), (320, 640, 1, 1)]:
    x = torch.randn(B, Cin, T, device="cuda")
    w = torch.randn(M, Cin, KS, device="cuda") / (Cin * KS) ** 0.5
    wp = H.pack_conv_fwd(w)
    for _ in range(3):
        st = H.conv_nn(x, wp, M, KS, dil, want_pre=True, want_out=False, want_stats=True)[2]
    torch.cuda.synchronize()
    v = st.flatten()[:32].cpu().view(4, 8)
    print(f"{Cin}->{M} k{KS} d{dil}: per stage cycles [first 60 MFMAs (+ window split) | barrier | last 30 MFMAs + next fragments | total]")
    for w_ in range(4):
        print("   wave", w_, [int(t) for t in v[w_, :4]], "| prologue", int(v[w_, 4]), "main", int(v[w_, 5]),
              "epilogue issued", int(v[w_, 6]), "drained", int(v[w_, 7]))
