import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import brainmagick_amd
from brainmagick_amd import hip_ops as H
from brainmagick_amd._lib import lib, check
brainmagick_amd.set_compute_dtype("f32x3")
B, T, M, Cin, KS = 256, 360, 320, 320, 3
for dil in (1, 16):
    x = torch.randn(B, Cin, T, device="cuda")
    dy = torch.randn(B, M, T, device="cuda")
    ns = lib().bm_gemm_nt_x3_suggest_splits(M, Cin, KS, B, T, 1, dil)
    part = torch.zeros(ns * M * Cin * KS, device="cuda")
    for _ in range(3):
        check(lib().bm_gemm_nt_x3(dy.data_ptr(), M * T, T, x.data_ptr(), Cin * T, T, None, None, part.data_ptr(),
                                  B, 1, M, Cin, T, KS, dil, ns, None), "gemm")
    torch.cuda.synchronize()
    v = part[:32].cpu().view(4, 8)
    print(f"dil {dil} nsplit {ns}: per stage [first 60 MFMAs + split | barrier | early fragments | vm wait | pieces 6-7 + fetch + last 30 MFMAs | total] stages")
    for w in range(4):
        print("   wave", w, [int(t) for t in v[w, :7]])
