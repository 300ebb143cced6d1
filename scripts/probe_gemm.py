"""Timing of the time-contraction GEMM (wgrad / ClipLoss scores shapes of the cfg2 step) per compute mode."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import brainmagick_amd
from brainmagick_amd import hip_ops as H

modes = sys.argv[1:] or ["f32x3"]
B, T = 256, 360
shapes = [(320, 320, 3, 1), (320, 320, 3, 2), (320, 320, 3, 16), (640, 320, 3, 1), (270, 270, 1, 1), (640, 320, 1, 1),
          (120, 640, 1, 1)]
for mode in modes:
    brainmagick_amd.set_compute_dtype(mode)
    for (M, Cin, KS, dil) in shapes:
        x = torch.randn(B, Cin, T, device="cuda")
        dy = torch.randn(B, M, T, device="cuda")
        for _ in range(3):
            H.gemm_nt(dy, x, B, M, Cin, T, KS, dil)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(10):
            H.gemm_nt(dy, x, B, M, Cin, T, KS, dil)
        en.record()
        torch.cuda.synchronize()
        ms = st.elapsed_time(en) / 10
        print(f"{mode:6s} wgrad M={M} C={Cin} k{KS} d{dil}: {ms * 1e3:8.1f} us  {2.0 * B * T * M * Cin * KS / ms / 1e9:7.1f} TF-eq", flush=True)
