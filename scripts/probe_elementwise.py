"""HIP-event timing of the streaming kernels at the paper shape (B=256, C=320, T=360)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from brainmagick_amd import hip_ops as H  # noqa: E402

B, C, T = 256, 320, 360
g = torch.Generator().manual_seed(0)
y = torch.randn(B, C, T, generator=g).cuda()
d = torch.randn(B, C, T, generator=g).cuda()
u = torch.randn(B, 2 * C, T, generator=g).cuda()
sc, sh = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
mean, invstd = torch.randn(C, generator=g).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


mb = B * C * T * 4 / 1e6
for name, fn, nbytes in (
        ("act_bn_bwd (reduce + apply)", lambda: H.act_bn_bwd(d, y, sc, sh, mean, invstd, True, H.ACT_GELU, want_affine_grads=True), 5 * mb),
        ("affine_act_res", lambda: H.affine_act_res(y, sc, sh, d, H.ACT_GELU), 3 * mb),
        ("channel_stats", lambda: H.channel_stats(y), mb),
        ("glu_fwd", lambda: H.glu_fwd(u), 3 * mb),
        ("glu_bwd", lambda: H.glu_bwd(d, u, want_dbias=True), 5 * mb)):
    t = timeit(fn)
    print(f"{name:30s} {t:7.1f} us  {nbytes / t:5.2f} TB/s", flush=True)        # MB / us = TB/s
