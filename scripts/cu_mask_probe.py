"""Does a CU-masked stream keep the MFMA kernels' throughput (the matrix pipe is power-limited) and what does an
HBM-bound kernel reach on the complementary CUs?  python scripts/cu_mask_probe.py"""
import ctypes
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import brainmagick_amd  # noqa: E402
from brainmagick_amd import hip_ops as H  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def timeit(fn, stream, reps=10):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


T, Cin, M, KS, dil = 360, 320, 320, 3, 2
g = torch.Generator().manual_seed(0)
w = (torch.randn(M, Cin, KS, generator=g) / math.sqrt(Cin * KS)).cuda()
brainmagick_amd.set_compute_dtype("f16x2")
wp = H.pack_conv_fwd(w, (T, dil))
full = (1 << 256) - 1
for name, bits, ncu in (("all 256", full, 256), ("first 192", (1 << 192) - 1, 192), ("3 of every 4", int("7" * 64, 16), 192),
                        ("first 128", (1 << 128) - 1, 128)):
    B = ncu                       # 2 time tiles per segment -> 2 workgroups per CU
    x = torch.randn(B, Cin, T, generator=g).cuda()
    H.amax(x)
    st = masked_stream(bits)
    t = timeit(lambda: H.conv_nn(x, wp, M, KS, dil, want_pre=True, want_out=False), st)
    fl = 2.0 * B * T * M * Cin * KS
    print(f"conv 320->320 k3 on {name:14s}: B={B} {t * 1e6:7.1f} us  {fl / t / 1e12:6.1f} TF  ({fl / t / 1e12 / ncu:5.2f} TF / CU)", flush=True)
    a = torch.randn(256, 320, T, generator=g).cuda()
    sc = torch.ones(320).cuda()
    sh = torch.zeros(320).cuda()
    t2 = timeit(lambda: H.affine_act_res(a, sc, sh, a, H.ACT_GELU, 0.0), st)
    print(f"   affine_act_res (354 MB) on the same CUs: {t2 * 1e6:7.1f} us  {0.354 / t2 / 1e3:5.2f} TB/s", flush=True)
# concurrency: conv on 192 CUs while the streaming kernel runs on the other 64
conv_st = masked_stream((1 << 192) - 1)
ew_st = masked_stream(((1 << 64) - 1) << 192)
x = torch.randn(192, Cin, T, generator=g).cuda()
H.amax(x)
a = torch.randn(256, 320, T, generator=g).cuda()
sc, sh = torch.ones(320).cuda(), torch.zeros(320).cuda()
t_conv = timeit(lambda: H.conv_nn(x, wp, M, KS, dil, want_pre=True, want_out=False), conv_st)
t_ew = timeit(lambda: H.affine_act_res(a, sc, sh, a, H.ACT_GELU, 0.0), ew_st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n_conv, n_ew = 20, max(1, int(20 * t_conv / t_ew))
e0.record()
conv_st.wait_event(e0)
ew_st.wait_event(e0)
with torch.cuda.stream(conv_st):
    for _ in range(n_conv):
        H.conv_nn(x, wp, M, KS, dil, want_pre=True, want_out=False)
    ec = torch.cuda.Event(enable_timing=True)
    ec.record()
with torch.cuda.stream(ew_st):
    for _ in range(n_ew):
        H.affine_act_res(a, sc, sh, a, H.ACT_GELU, 0.0)
    ee = torch.cuda.Event(enable_timing=True)
    ee.record()
torch.cuda.synchronize()
print(f"alone: conv(192 CUs) {t_conv * 1e6:.1f} us, affine(64 CUs) {t_ew * 1e6:.1f} us; concurrent: {n_conv} convs in "
      f"{e0.elapsed_time(ec) / n_conv * 1e3:.1f} us each, {n_ew} affines in {e0.elapsed_time(ee) / n_ew * 1e3:.1f} us each")
