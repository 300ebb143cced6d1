"""Timing of the implicit-GEMM conv at the cfg2 layer shapes, per compute mode (HIP events)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import brainmagick_amd
from brainmagick_amd import hip_ops as H

modes = sys.argv[1:] or ["f32x3"]
B, T = 256, 360
shapes = [(320, 320, 3, 1), (320, 320, 3, 2), (320, 320, 3, 16), (320, 640, 3, 1), (320, 640, 1, 1), (640, 320, 1, 1)]
for mode in modes:
    brainmagick_amd.set_compute_dtype(mode)
    for (Cin, M, KS, dil) in shapes:
        x = torch.randn(B, Cin, T, device="cuda")
        w = torch.randn(M, Cin, KS, device="cuda") / (Cin * KS) ** 0.5
        b = torch.randn(M, device="cuda")
        wp = H.pack_conv_fwd(w)
        for _ in range(3):
            H.conv_nn(x, wp, M, KS, dil, bias=b, want_pre=True, want_out=False)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(10):
            H.conv_nn(x, wp, M, KS, dil, bias=b, want_pre=True, want_out=False)
        en.record()
        torch.cuda.synchronize()
        ms = st.elapsed_time(en) / 10
        print(f"{mode:6s} conv {Cin}->{M} k{KS} d{dil}: {ms * 1e3:8.1f} us  {2.0 * B * T * M * Cin * KS / ms / 1e9:7.1f} TF-eq", flush=True)
