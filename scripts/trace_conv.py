"""Cycle trace of the wide f16x2 conv kernel (diagnostic library built by scripts/build_trace_lib.sh):
BM_HIP_LIB=brainmagick_amd/libbmhip_trace.so python scripts/trace_conv.py"""
import ctypes
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import brainmagick_amd  # noqa: E402
from brainmagick_amd import hip_ops as H  # noqa: E402

B, T = 256, 360
# (the production main loop stamps a stage as one segment; the LDS-DMA loop of conv_nn_h2d.hip had five)
SEG = ["stage", "-", "-", "-", "-"]
g = torch.Generator().manual_seed(0)
# what the epilogue carries: y_pre alone, y_pre + BatchNorm partial sums (the forward convs of the stack), y_out +
# residual (the data-gradient convs of the residual layers)
for Cin, M, KS, dil, what in [(320, 320, 3, 2, "pre"), (320, 320, 3, 2, "pre+stats"), (320, 320, 3, 2, "out+res"),
                              (320, 640, 1, 1, "pre")]:
    x = torch.randn(B, Cin, T, generator=g).cuda()
    w = (torch.randn(M, Cin, KS, generator=g) / math.sqrt(Cin * KS)).cuda()
    bias = torch.randn(M, generator=g).cuda()
    res = torch.randn(B, M, T, generator=g).cuda() if what == "out+res" else None
    brainmagick_amd.set_compute_dtype("f16x2")
    wp = H.pack_conv_fwd(w, (T, dil))
    for _ in range(3):
        if what == "out+res":
            H.conv_nn(x, wp, M, KS, dil, bias=bias, res=res)
        else:
            H.conv_nn(x, wp, M, KS, dil, bias=bias, want_pre=True, want_out=False, want_stats=what == "pre+stats")
    torch.cuda.synchronize()
    out = (ctypes.c_uint * (64 * 4 * 24))()
    rc = H.lib().bm_debug_trace_read_conv(out)
    assert rc == 0, rc
    tr = torch.tensor(list(out), dtype=torch.float64).view(64, 4, 3, 8)
    print(f"conv Cin={Cin} M={M} KS={KS} [{what}]: per tile, cycles: prologue {tr[..., 0, 5].mean().item():.0f}, main loop "
          f"{tr[..., 1, 5].mean().item():.0f}, last term + epilogue (stores drained) {tr[..., 0, 6].mean().item():.0f}")
    e = tr.mean((0, 1))
    print(f"  epilogue segments: staging {e[0, 1]:.0f}, finish accumulators {e[0, 2]:.0f}, BatchNorm sums {e[0, 3]:.0f}, "
          f"stores issued {e[1, 1]:.0f}, amax {e[1, 2]:.0f}")
    for j in range(KS):
        n = tr[..., j, 7].clamp(min=1)
        per = tr[..., j, :5] / n[..., None]
        print(f"  tap {j}: stages {n[0, 0].item():.0f}, cycles/stage {per.sum(-1).mean().item():.0f}: " + "; ".join(
            f"{name} {per[..., i].mean().item():.0f}" for i, name in enumerate(SEG)))
