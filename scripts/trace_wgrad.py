"""Cycle trace of the wide weight-gradient kernel (diagnostic library built by scripts/build_trace_lib.sh):
BM_HIP_LIB=brainmagick_amd/libbmhip_trace.so python scripts/trace_wgrad.py"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import brainmagick_amd  # noqa: E402
from brainmagick_amd import hip_ops as H  # noqa: E402

B, T = 256, 360
SEG = ["wait loads", "k-step 0 + split", "fetch issue", "k-step 1 (30 MFMA)", "barrier", "early frags + 15 MFMA"]
g = torch.Generator().manual_seed(0)
import os
SHAPES = [(320, 320, 3, 2), (320, 640, 1, 1), (256, 256, 0, 120), (256, 256, 0, 1024)]
if os.environ.get("TRACE_ONLY_CLIP"):
    SHAPES = SHAPES[2:]
for Cin, M, KS, dil in SHAPES:
    brainmagick_amd.set_compute_dtype("f16x2")
    if KS == 0:                 # the ClipLoss score contraction at F = dil: one segment of K = F * T samples
        K = dil * T
        est = torch.randn(256, K, generator=g).cuda()
        cand = torch.randn(256, K, generator=g).cuda()
        for _ in range(3):
            H.gemm_nt_partials(est, cand, 1, 256, 256, K, (0, K), (0, K))
    else:
        x = torch.randn(B, Cin, T, generator=g).cuda()
        dy = torch.randn(B, M, T, generator=g).cuda()
        for _ in range(3):
            H.gemm_nt(dy, x, B, M, Cin, T, KS, dil)
    torch.cuda.synchronize()
    out = (ctypes.c_longlong * (64 * 4 * 8))()
    rc = H.lib().bm_debug_trace_read(out)
    assert rc == 0, rc
    tr = torch.tensor(list(out), dtype=torch.float64).view(64, 4, 8)
    n = tr[..., 7].clamp(min=1)
    per = tr[..., :6].clone() / n[..., None]
    per[..., 2] = 0
    per[..., 3] = 0
    print(f"wgrad Cin={Cin} M={M} KS={KS}: stages/workgroup {n[0, 0].item():.0f}; cycles per stage "
          f"{(tr[..., 0] + tr[..., 1] + tr[..., 4] + tr[..., 5]).div(n).mean().item():.0f}; per workgroup: prologue "
          f"{tr[..., 2].mean().item():.0f}, main loop {tr[..., 3].mean().item():.0f}, partial-tile store "
          f"{tr[..., 6].mean().item():.0f} cycles")
    for i, name in enumerate(SEG):
        print(f"  {name:24s} mean {per[..., i].mean().item():8.0f}  min {per[..., i].min().item():8.0f}  "
              f"max {per[..., i].max().item():8.0f}")
    print("  per wave of workgroup 0:", [[round(v) for v in per[0, w].tolist()] for w in range(4)])
