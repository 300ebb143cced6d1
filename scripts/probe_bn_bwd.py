"""HIP-event time of bm_act_bn_bwd (train-mode BatchNorm + GELU backward of one 320-channel layer at B = 256, T = 360)
in its two forms, interleaved: 0 = two passes, 1 = one pass (csrc/norm_act.hip, bn_bwd_fused_kernel)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from brainmagick_amd import hip_ops as H          # noqa: E402
from brainmagick_amd._lib import lib              # noqa: E402

B, C, T = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 320, 360)
g = torch.Generator().manual_seed(0)
y = (torch.randn(B, C, T, generator=g) * 1.5).cuda()
dout = torch.randn(B, C, T, generator=g).cuda()
mean = y.mean((0, 2)); invstd = 1.0 / torch.sqrt(y.var((0, 2), unbiased=False) + 1e-5)
scale = (torch.rand(C, generator=g).cuda() + 0.5) * invstd
shift = -mean * scale
other = torch.randn(64 << 20, device="cuda")           # 256 MB streamed between calls: nothing of a call stays cached
res = {}
for rnd in range(3):
    for mode in (0, 1):
        lib().bm_act_bn_bwd_set_fused(mode)
        ts = []
        for rep in range(12):
            other.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            H.act_bn_bwd(dout, y, scale, shift, mean, invstd, True, H.ACT_GELU, want_affine_grads=True)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res.setdefault(mode, []).append(sorted(ts)[len(ts) // 2])
nbytes = {0: 5, 1: 3}
for mode, v in res.items():
    us = sorted(v)[1]
    print(f"mode {mode}: median {us:7.1f} us per call (rounds {[round(x, 1) for x in v]}), "
          f"{nbytes[mode] * B * C * T * 4 / us / 1e6:.2f} TB/s of algorithmic traffic")
