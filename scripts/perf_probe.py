"""Kernel-level timing probe at cfg2/cfg3 layer shapes (HIP events, on the current stream)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from brainmagick_amd import hip_ops as H  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters


def main():
    B, T = 256, 360
    dev = "cuda"
    print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count)
    rows = []
    for (Cin, M, KS, dil) in [(320, 320, 3, 1), (320, 320, 3, 16), (320, 640, 3, 1), (270, 270, 1, 1),
                              (208, 270, 1, 1), (320, 640, 1, 1), (640, 120, 1, 1), (640, 1024, 1, 1)]:
        x = torch.randn(B, Cin, T, device=dev)
        w = torch.randn(M, Cin, KS, device=dev) / (Cin * KS) ** 0.5
        b = torch.randn(M, device=dev)
        wp = H.pack_conv_fwd(w)
        flops = 2.0 * B * T * M * Cin * KS
        ms = timeit(lambda: H.conv_nn(x, wp, M, KS, dil, bias=b, want_pre=True, want_out=False,
                                      want_stats=True))
        rows.append((f"conv_nn fwd+stats {Cin}->{M} k{KS} d{dil}", ms, flops / ms / 1e9))
        dy = torch.randn(B, M, T, device=dev)
        wpd = H.pack_conv_dgrad(w)
        ms = timeit(lambda: H.conv_nn(dy, wpd, Cin, KS, dil))
        rows.append((f"conv_nn dgrad    {M}->{Cin} k{KS} d{dil}", ms, flops / ms / 1e9))
        ms = timeit(lambda: H.gemm_nt(dy, x, B, M, Cin, T, KS, dil))
        rows.append((f"gemm_nt wgrad    {M}x{Cin} k{KS} d{dil}", ms, flops / ms / 1e9))
        ms = timeit(lambda: H.pack_conv_fwd(w))
        rows.append((f"pack fwd         {M}x{Cin}x{KS}", ms, 0))
        ms = timeit(lambda: H.pack_conv_dgrad(w))
        rows.append((f"pack dgrad       {M}x{Cin}x{KS}", ms, 0))
    # elementwise
    y = torch.randn(B, 320, T, device=dev)
    res = torch.randn(B, 320, T, device=dev)
    sc = torch.rand(320, device=dev) + 0.5
    sh = torch.randn(320, device=dev)
    nbytes = y.numel() * 4
    ms = timeit(lambda: H.affine_act_res(y, sc, sh, res, H.ACT_GELU))
    rows.append(("affine_act_res 320", ms, 3 * nbytes / ms / 1e6))
    mean = torch.zeros(320, device=dev)
    ms = timeit(lambda: H.act_bn_bwd(res, y, sc, sh, mean, sc, True, H.ACT_GELU, want_affine_grads=True))
    rows.append(("act_bn_bwd 320 (GB/s of 5 passes)", ms, 5 * nbytes / ms / 1e6))
    u = torch.randn(B, 640, T, device=dev)
    ms = timeit(lambda: H.glu_fwd(u))
    rows.append(("glu_fwd 640", ms, 3 * nbytes / ms / 1e6))
    ms = timeit(lambda: H.glu_bwd(y, u))
    rows.append(("glu_bwd 640", ms, 5 * nbytes / ms / 1e6))
    ms = timeit(lambda: H.channel_sum(y))
    rows.append(("channel_sum 320", ms, nbytes / ms / 1e6))
    # clip
    for Fd in (120, 1024):
        K = Fd * T
        est = torch.randn(B, Fd, T, device=dev)
        cand = torch.randn(B, Fd, T, device=dev)
        flops = 2.0 * B * B * K
        ms = timeit(lambda: H.clip_inv_norms(cand))
        rows.append((f"clip_inv_norms F={Fd} (GB/s)", ms, cand.numel() * 4 / ms / 1e6))
        ms = timeit(lambda: H.gemm_nt_partials(est, cand, 1, B, B, K, (0, K), (0, K)))
        rows.append((f"clip scores gemm_nt F={Fd}", ms, flops / ms / 1e9))
        ds = torch.randn(B, B, device=dev)
        wp = H.pack_weights(ds, 1, B, B, 1, 0, B, 1, 0)
        ms = timeit(lambda: H.conv_nn(cand.view(1, B, K), wp, B, 1, 1))
        rows.append((f"clip dEst conv_nn F={Fd}", ms, flops / ms / 1e9))
    for name, ms, rate in rows:
        print(f"{name:45s} {ms:9.3f} ms   {rate:10.1f} GFLOP/s|GB/s")


if __name__ == "__main__":
    main()
