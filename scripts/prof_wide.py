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
        _, _, st = H.conv_nn(x, wp, M, KS, dil, want_pre=True, want_out=False, want_stats=True)
    torch.cuda.synchronize()
    raw = st.flatten()[:32].cpu().view(4, 8)
    v = raw[:, :5]
    tot = st.flatten()[32:36].cpu()
    print(f"   prologue {float(raw[0, 7]):.0f} cycles, main loop {float(raw[0, 5]):.0f}, whole block (stores drained) {float(tot[0]):.0f}")
    print(f"   shader clock during the main loop: {float(raw[0, 5]) / float(raw[0, 6]) * 100:.0f} MHz "
          f"({float(raw[0, 5]):.0f} cycles, {float(raw[0, 6]) / 100:.1f} us)")
    print(f"{Cin}->{M} k{KS} d{dil}: per-stage cycles [issue dma/x | frag reads | mfma issue | x split+store | barrier]")
    for w_ in range(4):
        print("   wave", w_, [round(float(t)) for t in v[w_]], "total", round(float(v[w_].sum())))
