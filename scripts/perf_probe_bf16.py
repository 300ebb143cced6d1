import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import brainmagick_amd
from brainmagick_amd import hip_ops as H
from perf_probe import timeit

def main():
    B, T, dev = 256, 360, "cuda"
    for mode in ("f32", "bf16"):
        brainmagick_amd.set_compute_dtype(mode)
        for (Cin, M, KS, dil) in [(320, 320, 3, 1), (320, 320, 3, 16), (320, 640, 3, 1), (270, 270, 1, 1), (320, 640, 1, 1), (640, 120, 1, 1)]:
            x = torch.randn(B, Cin, T, device=dev)
            w = torch.randn(M, Cin, KS, device=dev) / (Cin * KS) ** 0.5
            b = torch.randn(M, device=dev)
            wp = H.pack_conv_fwd(w)
            flops = 2.0 * B * T * M * Cin * KS
            ms = timeit(lambda: H.conv_nn(x, wp, M, KS, dil, bias=b, want_pre=True, want_out=False), iters=20, warm=3)
            gb = (x.numel() + B * M * T) * 4 / 1e9
            print(f"{mode} conv_nn {Cin}->{M} k{KS} d{dil}: {ms:.3f} ms  {flops/ms/1e9:.1f} TF  {gb/ms*1e3:.0f} GB/s algorithmic")
            ms = timeit(lambda: H.pack_conv_fwd(w))
            print(f"   pack {ms:.3f} ms")
main()
