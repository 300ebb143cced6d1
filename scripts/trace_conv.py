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
SEG = ["early frags + 15 MFMA (prev)", "issue: X loads, DMA, late frags", "wait X window", "split+store / 30 MFMA", "barrier"]
g = torch.Generator().manual_seed(0)
for Cin, M, KS, dil in [(320, 320, 3, 2), (320, 640, 1, 1)]:
    x = torch.randn(B, Cin, T, generator=g).cuda()
    w = (torch.randn(M, Cin, KS, generator=g) / math.sqrt(Cin * KS)).cuda()
    brainmagick_amd.set_compute_dtype("f16x2")
    wp = H.pack_conv_fwd(w, (T, dil))
    for _ in range(3):
        H.conv_nn(x, wp, M, KS, dil, want_pre=True, want_out=False)
    torch.cuda.synchronize()
    out = (ctypes.c_uint * (64 * 4 * 24))()
    rc = H.lib().bm_debug_trace_read_conv(out)
    assert rc == 0, rc
    tr = torch.tensor(list(out), dtype=torch.float64).view(64, 4, 3, 8)
    print(f"conv Cin={Cin} M={M} KS={KS}: per tile, cycles: prologue {tr[..., 0, 5].mean().item():.0f}, main loop "
          f"{tr[..., 1, 5].mean().item():.0f}, last term + epilogue (stores drained) {tr[..., 0, 6].mean().item():.0f}")
    for j in range(KS):
        n = tr[..., j, 7].clamp(min=1)
        per = tr[..., j, :5] / n[..., None]
        print(f"  tap {j}: stages {n[0, 0].item():.0f}, cycles/stage {per.sum(-1).mean().item():.0f}: " + "; ".join(
            f"{name} {per[..., i].mean().item():.0f}" for i, name in enumerate(SEG)))
