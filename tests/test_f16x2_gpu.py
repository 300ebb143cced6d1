"""Compute mode "f16x2": fp32-accurate contractions on the f16 matrix cores (per-tensor / per-row power-of-two
scaling, two f16 planes per operand, three partial products, fp32 accumulate -- csrc/conv_nn_h2w.hip,
csrc/gemm_nt_h2w.hip).  This is the library default: the golden-vector, oracle, full-size and
training-curve suites of test_model_gpu.py run in it, at the SAME tolerances as the exact-fp32 MFMA mode (per-kernel
forward rel-L2 <= 5e-6 vs fp64, gradients <= 2e-5, end-to-end 1e-5 / 1e-4 / loss 1e-4).  This file holds the
mode-specific kernel tests: every tile variant of the wide kernels, the scaling machinery (dynamic range, per-row
weight scales, non-finite inputs), and the error vs fp64 next to the other fp32-class modes'."""
import math

import pytest
import torch
from torch.nn import functional as F

import test_kernels_gpu as TK
from helpers import rel_l2

pytestmark = pytest.mark.gpu

FWD_TOL = TK.FWD_TOL
GRAD_TOL = TK.GRAD_TOL


@pytest.fixture()
def h2_mode():
    import brainmagick_amd
    default = brainmagick_amd.get_compute_dtype()
    brainmagick_amd.set_compute_dtype("f16x2")
    yield
    brainmagick_amd.set_compute_dtype(default)


@pytest.fixture(scope="module")
def H():
    from brainmagick_amd import hip_ops
    return hip_ops


def _gen(seed):
    return torch.Generator().manual_seed(seed)


# Cin, M, KS, dil, T, B -- every MW variant of the wide kernel (M -> 64 MW row tiles, MW in {5, 4, 2}), padded
# rows / channels, both kernel sizes, tile-boundary lengths, plus the generic cases (some fall to the narrow kernels)
H2_CASES = TK.CONV_CASES + [
    (270, 270, 1, 1, 360, 2),      # the 270-channel front end: MW = 5, 50 padded rows
    (208, 270, 1, 1, 361, 2),      # merger apply
    (270, 208, 1, 1, 343, 2),      # its data gradient: MW = 4 (256 rows)
    (640, 120, 1, 1, 360, 2),      # mel head: MW = 2 (128 rows)
    (640, 1024, 1, 1, 200, 1),     # wav2vec head: MW = 4, four row tiles
    (256, 256, 1, 1, 1200, 1),     # ClipLoss dEst shape (long time axis, one segment)
    (320, 320, 3, 8, 193, 2),      # second time tile holds one column
    (24, 128, 3, 1, 130, 3),       # MW = 2 with 3 taps
]


def _covered(H, Cin, M, T, KS, dil):
    return bool(H.lib().bm_conv_h2_covers(Cin, M, T, KS, dil))


@pytest.mark.parametrize("Cin,M,KS,dil,T,B", H2_CASES)
def test_conv_forward(h2_mode, H, Cin, M, KS, dil, T, B):
    g = _gen(Cin * 7 + M + KS + dil + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(M, Cin, KS, generator=g) / math.sqrt(Cin * KS)
    b = torch.randn(M, generator=g)
    ref = F.conv1d(x.double(), w.double(), b.double(), padding=KS // 2 * dil, dilation=dil)
    wp = H.pack_conv_fwd(w.cuda(), (T, dil))
    assert wp._bm_mode == ("f16x2" if _covered(H, Cin, M, T, KS, dil) else "f32x3")
    _, y, _ = H.conv_nn(x.cuda(), wp, M, KS, dil, bias=b.cuda())
    assert y.shape == ref.shape
    assert rel_l2(y, ref) < FWD_TOL


def test_the_lds_dma_main_loop_kept_for_ab_runs_still_passes():
    """`BM_CONV_LDSDMA=1` selects the round-2..5 main loop of the wide conv (conv_nn_h2d.hip: weight slabs by LDS-DMA), which
    ships for interleaved A/B runs; the switch is read once per process, so the conv tests (forward shapes, epilogues,
    BatchNorm sums, backward) are re-run in a child process under it."""
    import os
    import subprocess
    import sys
    if os.environ.get("BM_CONV_LDSDMA") == "1":
        pytest.skip("already the child")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                          "test_conv_forward or test_conv_epilogues or test_conv_epilogue_batchnorm_statistics or "
                          "test_conv_backward"],
                         env=dict(os.environ, BM_CONV_LDSDMA="1"), cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and " failed" not in out.stdout, out.stdout[-500:]


def test_wide_kernel_is_what_runs_for_the_paper_shapes(H, h2_mode):
    for Cin, M, KS, dil, T in [(270, 320, 3, 1, 360), (320, 320, 3, 16, 343), (320, 640, 3, 1, 361),
                               (320, 640, 1, 1, 360), (640, 120, 1, 1, 360), (208, 270, 1, 1, 360),
                               (270, 270, 1, 1, 360), (256, 256, 1, 1, 43200)]:
        assert _covered(H, Cin, M, T, KS, dil), (Cin, M, KS, dil, T)
    assert not _covered(H, 320, 320, 3, 32, 360)       # halo beyond the staged window
    assert not _covered(H, 16, 16, 3, 4, 7)


@pytest.mark.parametrize("M,KS", [(320, 3), (256, 1), (128, 3), (270, 1)])
def test_conv_epilogues(h2_mode, H, M, KS):
    """bias / pre-activation store / per-channel affine / exact-erf GELU / residual in the wide kernel's general
    epilogue, against fp64."""
    g = _gen(M + KS)
    B, Cin, T, dil = 2, 64, 300, 2
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(M, Cin, KS, generator=g) / math.sqrt(Cin * KS)
    b = torch.randn(M, generator=g)
    scale = torch.rand(M, generator=g) + 0.5
    shift = torch.randn(M, generator=g)
    res = torch.randn(B, M, T, generator=g)
    pre_ref = F.conv1d(x.double(), w.double(), b.double(), padding=KS // 2 * dil, dilation=dil)
    out_ref = F.gelu(pre_ref * scale.double()[None, :, None] + shift.double()[None, :, None]) + res.double()
    wp = H.pack_conv_fwd(w.cuda(), (T, dil))
    assert wp._bm_mode == "f16x2"
    pre, out, _ = H.conv_nn(x.cuda(), wp, M, KS, dil, bias=b.cuda(), scale=scale.cuda(), shift=shift.cuda(),
                            res=res.cuda(), act=H.ACT_GELU, want_pre=True)
    assert rel_l2(pre, pre_ref) < FWD_TOL and rel_l2(out, out_ref) < FWD_TOL
    # the simple epilogue (one output, optional residual)
    _, out2, _ = H.conv_nn(x.cuda(), wp, M, KS, dil, bias=b.cuda(), res=res.cuda())
    assert rel_l2(out2, pre_ref + res.double()) < FWD_TOL
    pre3, none, _ = H.conv_nn(x.cuda(), wp, M, KS, dil, want_pre=True, want_out=False)
    assert none is None and rel_l2(pre3, pre_ref - b.double()[None, :, None]) < FWD_TOL


def test_amax_kernel(H):
    g = _gen(1)
    for n in (1, 5, 1024, 4 * 1000 + 3, 3 * 208 * 360):
        x = torch.randn(n, generator=g) * 3
        got = H.amax(x.cuda())
        assert float(got.max()) == float(x.abs().max())
    x = torch.zeros(64).cuda()
    assert float(H.amax(x).max()) == 0.0
    # a tensor that is only 4-byte aligned (a batch slice such as meg[1:] with C * T odd): scalar head up to the first
    # 16-byte boundary, vector body, scalar tail
    for n, off in ((4099, 1), (4099, 3), (7, 2), (2, 1)):
        base = torch.randn(n + 8, generator=g).cuda() * 2
        view = base[off:off + n]
        assert view.data_ptr() % 16 != 0 and view.is_contiguous()
        assert float(H.amax(view).max()) == float(view.abs().max()), (n, off)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    bad = torch.randn(4099, generator=g).cuda()
    bad[1] = float("inf")                       # inside the scalar head of the misaligned view below
    H.amax(bad[1:], nonfinite_flag=flag)
    assert int(flag) == 1
    # cached per tensor version: an in-place change invalidates the cache
    y = torch.ones(1000).cuda()
    a1 = H.amax(y)
    assert H.amax(y) is a1
    y.mul_(5)
    assert float(H.amax(y).max()) == 5.0


@pytest.mark.parametrize("xs,ws", [(1e-6, 1.0), (300.0, 1e-4), (1.0, 1e5), (2 ** -40, 2 ** 30)])
def test_dynamic_range_is_handled_by_the_scales(h2_mode, H, xs, ws):
    """f16 has 5 exponent bits: tensors far from 1 in magnitude (tiny gradients, large weights) must come out
    with the same relative accuracy, and weight rows of very different magnitude (per-row scale) too."""
    g = _gen(7)
    B, Cin, M, KS, dil, T = 2, 96, 320, 3, 1, 256
    x = torch.randn(B, Cin, T, generator=g) * xs
    w = torch.randn(M, Cin, KS, generator=g) / math.sqrt(Cin * KS) * ws
    w = w * torch.logspace(-6, 6, M)[:, None, None]          # rows spanning 12 decades
    ref = F.conv1d(x.double(), w.double(), None, padding=dil, dilation=dil)
    _, y, _ = H.conv_nn(x.cuda(), H.pack_conv_fwd(w.cuda(), (T, dil)), M, KS, dil)
    # per output row (every row has its own scale): relative error at the fp32 level
    err = (y.double().cpu() - ref).norm(dim=(0, 2)) / ref.norm(dim=(0, 2))
    assert err.max().item() < FWD_TOL, err.max().item()


# ---- per-channel dynamic range INSIDE one activation / gradient tensor ----------------------------------------
# Activations carry ONE power-of-two scale per tensor (their maximum lands in [2^14, 2^15) of the f16 range).  A value
# v keeps 22 significant bits while |v| >= 2^-16 max|tensor|; below that its low plane sinks into the f16
# subnormals (spacing 2^-24 in scaled units) and the ABSOLUTE error stays <= 2^-39 max|tensor|.
#  * conv: a channel-wise relative bound holds whenever an output channel is not fed EXCLUSIVELY by inputs that are
#    more than 2^-16 below the tensor maximum: |err(y[m])| <= 2^-38 max|x| sum|w[m]| + the fp32 accumulation error;
#  * weight gradient: dW row m only sees dy channel m (and column c only x channel c), so with ONE scale per tensor a
#    channel d decades below its tensor's maximum has row-wise relative error <= max(fp32-class, 2^-37 * 10^d); rows
#    within 2^-16 of the maximum -- 4.8 decades -- are fp32-class.  The gradients that reach the weight-gradient kernel
#    from the BatchNorm / GLU backward kernels carry PER-CHANNEL maxima: then every row has its own scale and is
#    fp32-class whatever the spread (case (b) below); the x side keeps the per-tensor bound.
H2_ABS = 2.0 ** -37


@pytest.mark.parametrize("decades", [0, 2, 4, 6, 8])
def test_per_channel_spread_weight_gradient_rows(h2_mode, H, decades):
    g = _gen(40 + decades)
    B, Cin, M, KS, dil, T = 4, 128, 320, 3, 2, 256
    sdy = torch.logspace(0, -decades, M)
    sx = torch.logspace(0, -decades, Cin)
    dy = torch.randn(B, M, T, generator=g) * sdy[None, :, None]
    x = torch.randn(B, Cin, T, generator=g) * sx[None, :, None]
    assert H.lib().bm_gemm_nt_h2_covers(M, Cin, KS, B, T, 1, dil, 0)
    ref = _wgrad_ref(dy, x, KS, dil)
    col_bound = torch.clamp(H2_ABS / sx.double(), min=2e-6)
    dyg, xg = dy.cuda(), x.cuda()
    # (a) one scale per tensor (a gradient tensor whose producer published no per-channel maxima)
    dw = H.gemm_nt(dyg, xg, B, M, Cin, T, KS, dil)[0].double().cpu()
    assert rel_l2(dw, ref) < 1e-6                                     # norm-wise: always fp32-class
    row_err = (dw - ref).norm(dim=(1, 2)) / ref.norm(dim=(1, 2))      # per dy channel
    col_err = (dw - ref).norm(dim=(0, 2)) / ref.norm(dim=(0, 2))      # per x channel
    row_bound = torch.clamp(H2_ABS / sdy.double(), min=2e-6)
    print(f"spread 1e-{decades}, tensor scale: worst dW row {row_err.max():.2e}, column {col_err.max():.2e} "
          f"(rows within 2^-16 of the maximum: {row_err[sdy >= 2.0 ** -16].max():.2e})")
    assert bool((row_err <= row_bound).all()), (row_err / row_bound).max().item()
    assert bool((col_err <= col_bound).all()), (col_err / col_bound).max().item()
    assert row_err[sdy >= 2.0 ** -16].max().item() < 2e-6 and col_err[sx >= 2.0 ** -16].max().item() < 2e-6
    # (b) per-channel maxima of dy, as act_bn_bwd / glu_bwd publish them: every ROW of dW is fp32-class, whatever the
    # spread between the channels (the column bound is unchanged: x keeps its per-tensor scale)
    dyg._bm_row_amax = (dyg._version, dyg.data_ptr(), dyg.abs().amax(dim=(0, 2)).contiguous())
    dw = H.gemm_nt(dyg, xg, B, M, Cin, T, KS, dil)[0].double().cpu()
    row_err = (dw - ref).norm(dim=(1, 2)) / ref.norm(dim=(1, 2))
    col_err = (dw - ref).norm(dim=(0, 2)) / ref.norm(dim=(0, 2))
    print(f"spread 1e-{decades}, row scales:   worst dW row {row_err.max():.2e}, column {col_err.max():.2e}")
    assert rel_l2(dw, ref) < 1e-6
    assert row_err.max().item() < 2e-6, row_err.max().item()
    # element-wise against the row's own magnitude for the x channels within 2^-16 of their maximum
    ok_cols = sx >= 2.0 ** -16
    el = ((dw - ref).abs().amax(dim=2)[:, ok_cols] / ref.abs().amax(dim=(1, 2))[:, None]).max().item()
    assert el < 4e-6, el
    assert bool((col_err <= col_bound).all()), (col_err / col_bound).max().item()


@pytest.mark.parametrize("KS,T", [(1, 360), (3, 361), (3, 130)])
def test_row_scaled_weight_gradient_shapes(h2_mode, H, KS, T):
    """The row-scaled kernels on a 1x1 layer, and the fall-back to the per-tensor scale when T % 4 != 0 / short T."""
    g = _gen(60 + KS + T)
    B, Cin, M, dil = 3, 320, 320, 1
    dy = torch.randn(B, M, T, generator=g) * torch.logspace(0, -5, M)[None, :, None]
    x = torch.randn(B, Cin, T, generator=g)
    dyg = dy.cuda()
    dyg._bm_row_amax = (dyg._version, dyg.data_ptr(), dyg.abs().amax(dim=(0, 2)).contiguous())
    dw = H.gemm_nt(dyg, x.cuda(), B, M, Cin, T, KS, dil)[0]
    ref = _wgrad_ref(dy, x, KS, dil)
    assert rel_l2(dw, ref) < GRAD_TOL
    if T % 4 == 0:
        row_err = (dw.double().cpu() - ref).norm(dim=(1, 2)) / ref.norm(dim=(1, 2))
        assert row_err.max().item() < 2e-6, row_err.max().item()


@pytest.mark.parametrize("B,T,dil,Cin,M", [(7, 360, 1, 320, 320), (5, 360, 16, 320, 320), (9, 100, 4, 320, 320),
                                            (13, 68, 2, 64, 320), (3, 360, 8, 640, 320), (33, 76, 16, 320, 270)])
def test_weight_gradient_over_a_flat_time_axis(h2_mode, H, B, T, dil, Cin, M):
    """Row-scaled 3-tap weight gradients with T % 32 != 0 walk the B segments as ONE axis of B * T samples (no padding
    of every segment to a multiple of 32): a stage then straddles segment boundaries, a lane's piece belongs to the
    segment its own local time says, and the samples a tap shift moves outside their segment are zeroed per value.
    Boundaries at every position of a stage (T = 68, 76, 100), the largest dilation, a total that is not a multiple
    of 32 (B * T % 32 != 0), more segments than splits -- against fp64, row-wise."""
    g = _gen(B * T + dil)
    dy = torch.randn(B, M, T, generator=g) * torch.logspace(0, -4, M)[None, :, None]
    x = torch.randn(B, Cin, T, generator=g)
    # make the segment edges count: large values in the first / last samples of every segment
    x[:, :, :2] *= 8
    x[:, :, -2:] *= 8
    dy[:, :, :1] *= 4
    dy[:, :, -1:] *= 4
    from brainmagick_amd._lib import lib
    assert lib().bm_gemm_nt_h2_covers(M, Cin, 3, B, T, 1, dil, 0)
    dyg = dy.cuda()
    dyg._bm_row_amax = (dyg._version, dyg.data_ptr(), dyg.abs().amax(dim=(0, 2)).contiguous())
    dw = H.gemm_nt(dyg, x.cuda(), B, M, Cin, T, 3, dil)[0]
    ref = _wgrad_ref(dy, x, 3, dil)
    assert rel_l2(dw, ref) < GRAD_TOL
    row_err = (dw.double().cpu() - ref).norm(dim=(1, 2)) / ref.norm(dim=(1, 2))
    assert row_err.max().item() < 2e-6, row_err.max().item()
    tap_err = (dw.double().cpu() - ref).norm(dim=(0, 1)) / ref.norm(dim=(0, 1))        # per tap: the shifted edges
    assert tap_err.max().item() < 2e-6, tap_err


@pytest.mark.parametrize("decades", [0, 4, 8])
def test_per_channel_spread_conv_output_channels(h2_mode, H, decades):
    g = _gen(50 + decades)
    B, Cin, M, KS, dil, T = 3, 128, 320, 3, 2, 256
    sx = torch.logspace(0, -decades, Cin)
    x = torch.randn(B, Cin, T, generator=g) * sx[None, :, None]
    w = torch.randn(M, Cin, KS, generator=g) / math.sqrt(Cin * KS)
    w[M // 2:, :Cin // 2] = 0                     # the second half of the outputs only reads the SMALL channels
    ref = F.conv1d(x.double(), w.double(), None, padding=dil, dilation=dil)
    _, y, _ = H.conv_nn(x.cuda(), H.pack_conv_fwd(w.cuda(), (T, dil)), M, KS, dil)
    err = (y.double().cpu() - ref).abs().amax(dim=(0, 2))                               # per output channel
    # absolute bound of the split + fp32-class relative term on the channel's own magnitude
    bound = 2.0 ** -38 * x.abs().max().double() * w.double().abs().sum(dim=(1, 2)) + 3e-6 * ref.abs().amax(dim=(0, 2))
    rel = (y.double().cpu() - ref).norm(dim=(0, 2)) / ref.norm(dim=(0, 2))
    print(f"spread 1e-{decades}: worst output channel rel-L2 {rel.max():.2e} (first half {rel[:M // 2].max():.2e})")
    assert bool((err <= bound).all()), (err / bound).max().item()
    assert rel[:M // 2].max().item() < FWD_TOL     # channels fed by the large inputs: plain fp32 tolerance


def test_nonfinite_inputs_do_not_turn_finite(h2_mode, H):
    g = _gen(3)
    x = torch.randn(1, 32, 200, generator=g)
    x[0, 3, 17] = float("inf")
    w = torch.randn(128, 32, 1, generator=g)
    _, y, _ = H.conv_nn(x.cuda(), H.pack_conv_fwd(w.cuda(), (200, 1)), 128, 1, 1)
    assert not torch.isfinite(y[0, :, 17]).any()


def test_error_is_fp32_class(H):
    """rel-L2 vs fp64 of the three fp32-class modes on the same conv / weight gradient (K = 960 and 92 160)."""
    import brainmagick_amd
    g = _gen(0)
    B, Cin, M, KS, dil, T = 4, 320, 320, 3, 2, 360
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(M, Cin, KS, generator=g) / math.sqrt(Cin * KS)
    dy = torch.randn(B, M, T, generator=g)
    ref = F.conv1d(x.double(), w.double(), None, padding=dil, dilation=dil)
    wg = torch.zeros(M, Cin, KS, dtype=torch.float64, requires_grad=True)
    F.conv1d(x.double(), wg, None, padding=dil, dilation=dil).backward(dy.double())
    default = brainmagick_amd.get_compute_dtype()
    errs = {}
    for mode in ("f32", "f32x3", "f16x2"):
        brainmagick_amd.set_compute_dtype(mode)
        try:
            _, y, _ = H.conv_nn(x.cuda(), H.pack_conv_fwd(w.cuda(), (T, dil)), M, KS, dil)
            dw = H.gemm_nt(dy.cuda(), x.cuda(), B, M, Cin, T, KS, dil)[0]
        finally:
            brainmagick_amd.set_compute_dtype(default)
        errs[mode] = (rel_l2(y, ref), rel_l2(dw, wg.grad))
    print("rel-L2 vs fp64 (conv fwd, wgrad):", errs)
    for k in range(2):
        assert errs["f16x2"][k] < 1e-6
        assert errs["f16x2"][k] < 3 * errs["f32"][k] + 1e-7


def _wgrad_ref(dy, x, KS, dil):
    """dW[m][c][j] = sum_{b,t} dy[b][m][t] x[b][c][t + (j - KS//2) dil]  (fp64)."""
    B, M, T = dy.shape
    w = torch.zeros(M, x.shape[1], KS, dtype=torch.float64, requires_grad=True)
    F.conv1d(x.double(), w, None, padding=KS // 2 * dil, dilation=dil).backward(dy.double())
    return w.grad


@pytest.mark.parametrize("Cin,M,KS,dil,T,B", [(270, 270, 1, 1, 360, 4), (320, 640, 1, 1, 343, 3),
                                              (640, 1024, 1, 1, 361, 3), (64, 320, 3, 16, 361, 3),
                                              (300, 320, 3, 5, 200, 6), (320, 640, 3, 1, 360, 3),
                                              (270, 320, 3, 1, 360, 4)])
def test_weight_gradient(h2_mode, H, Cin, M, KS, dil, T, B):
    g = _gen(Cin + M + KS + T)
    x = torch.randn(B, Cin, T, generator=g)
    dy = torch.randn(B, M, T, generator=g)
    assert H.lib().bm_gemm_nt_h2_covers(M, Cin, KS, B, T, 1, dil, 0)
    dw = H.gemm_nt(dy.cuda(), x.cuda(), B, M, Cin, T, KS, dil)[0]
    assert rel_l2(dw, _wgrad_ref(dy, x, KS, dil)) < GRAD_TOL
    # tiny gradients / large activations: the per-tensor scales keep the relative accuracy
    dw2 = H.gemm_nt((dy * 1e-9).cuda(), (x * 300).cuda(), B, M, Cin, T, KS, dil)[0]
    assert rel_l2(dw2, _wgrad_ref(dy * 1e-9, x * 300, KS, dil)) < GRAD_TOL


@pytest.mark.parametrize("Cin,M,KS,dil,T,B", TK.CONV_CASES[:7] + [(320, 320, 3, 2, 360, 4), (128, 640, 3, 1, 361, 3)])
def test_conv_backward(h2_mode, H, Cin, M, KS, dil, T, B):
    TK.test_conv_backward_kernels(H, Cin, M, KS, dil, T, B)


def test_subject_layers_and_merger(h2_mode, H):
    TK.test_subject_layers_kernels(H)
    TK.test_merger_kernels(H)


@pytest.mark.parametrize("B,Bc,Fd,T", [(6, 6, 10, 48), (5, 12, 7, 33), (64, 64, 120, 360)])
def test_clip(h2_mode, H, B, Bc, Fd, T):
    TK.test_clip_kernels(H, B, Bc, Fd, T)


@pytest.mark.parametrize("B,Bc,K", [(256, 256, 43200), (256, 256, 368640), (250, 252, 4999), (256, 512, 1537),
                                    (180, 256, 777), (100, 256, 3000), (256, 2048, 20000)])
def test_score_contraction_256_tiles(h2_mode, H, B, Bc, K):
    """bm_clip_scores_h2 (256 x 256 tiles, candidates as the row operand, transposed 16-byte stores of the partial
    tiles): rows / columns past the operands, a ragged last chunk, K not a multiple of 4 (rows only 4-byte aligned),
    several row tiles -- against fp64, and against the generic 256 x 128 kernel it replaces."""
    from brainmagick_amd._lib import lib
    assert lib().bm_clip_scores_h2_covers(B, Bc, K)
    g = _gen(B + Bc + K)
    est = torch.randn(B, K, generator=g)
    cand = torch.randn(Bc, K, generator=g) * torch.logspace(0, -3, Bc)[:, None]     # candidates of different norm
    eg, cg = est.cuda(), cand.cuda()
    part = H.gemm_nt_partials(eg, cg, 1, B, Bc, K, (0, K), (0, K))
    assert part.shape[1:] == (B, Bc)
    got = part.sum(0).double().cpu()
    ref = est.double() @ cand.double().t()
    # norm-wise per candidate (column): every candidate keeps fp32-class accuracy relative to its own scores' scale...
    col_err = (got - ref).norm(dim=0) / ref.norm(dim=0)
    # ... down to the f16x2 bound of a candidate 2^-10 below the tensor maximum
    assert float(col_err.max()) < 2e-5, float(col_err.max())
    assert float(col_err[: Bc // 3].max()) < 2e-6, float(col_err[: Bc // 3].max())
    # the generic tiles agree (different split counts: fp32 summation order differs)
    if lib().bm_gemm_nt_h2_covers(B, Bc, 1, 1, K, 1, 1, 0):
        nsplit = lib().bm_gemm_nt_h2_suggest_splits(B, Bc, 1, 1, K)
        a_amax, x_amax = H.amax(eg), H.amax(cg)
        part2 = torch.empty(nsplit, B, Bc, device="cuda")
        from brainmagick_amd.hip_ops import check, _p, _stream
        check(lib().bm_gemm_nt_h2(_p(eg), 0, K, _p(a_amax), _p(cg), 0, K, _p(x_amax), _p(part2), 1, B, Bc, K, 1, 1,
                                  nsplit, _stream()), "bm_gemm_nt_h2")
        got2 = part2.sum(0).double().cpu()
        assert float((got - got2).norm() / ref.norm()) < 2e-6


@pytest.mark.parametrize("Cin,M,KS,dil,T,B", [(320, 320, 3, 2, 360, 3), (270, 320, 3, 1, 361, 2), (320, 270, 3, 4, 343, 2),
                                               (64, 640, 3, 16, 193, 2), (320, 320, 1, 1, 360, 2)])
def test_conv_epilogue_batchnorm_statistics(h2_mode, H, Cin, M, KS, dil, T, B):
    """The wide conv's epilogue writes the per-(tile, wavefront column) sums and sums of squares of y_pre that
    BatchNorm needs (every row of the stats buffer, also for tiles that straddle M or T); folded over the tiles
    they equal the sums of the stored output, and bn_finalize on them equals bn_finalize on a channel_stats pass."""
    g = _gen(Cin + M + T)
    x = torch.randn(B, Cin, T, generator=g).cuda()
    w = (torch.randn(M, Cin, KS, generator=g) / (Cin * KS) ** 0.5).cuda()
    bias = (torch.randn(M, generator=g) * 3).cuda()           # a mean far from zero in some channels
    wp = H.pack_conv_fwd(w, (T, dil))
    assert wp._bm_mode == "f16x2"
    pre, _, stats = H.conv_nn(x, wp, M, KS, dil, bias=bias, want_pre=True, want_out=False, want_stats=True)
    pre2, _, _ = H.conv_nn(x, wp, M, KS, dil, bias=bias, want_pre=True, want_out=False)
    assert torch.equal(pre, pre2)                             # the statistics do not disturb the output
    assert torch.isfinite(stats).all()                        # no row of the buffer left unwritten
    assert stats.shape[0] == M and stats._bm_channel_major         # [M][tiles][2]
    got = stats.double().sum(1).cpu()
    p64 = pre.double().cpu()
    ref = torch.stack([p64.sum(dim=(0, 2)), (p64 * p64).sum(dim=(0, 2))], dim=1)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-6
    assert float(((got[:, 1] - ref[:, 1]).abs() / ref[:, 1]).max()) < 2e-6
    a = H.bn_finalize(stats, B * T, None, None, None, None, None, 0.1, 1e-5)
    b = H.bn_finalize(H.channel_stats(pre), B * T, None, None, None, None, None, 0.1, 1e-5)
    # var = E[x^2] - mean^2 from fp32 partial sums: with a channel mean of 3 sigma both routes carry ~1e-6 * 10
    for u, v in zip(a, b):
        assert float((u - v).abs().max() / v.abs().max()) < 2e-5
    var64 = ref[:, 1] / (B * T) - (ref[:, 0] / (B * T)) ** 2
    assert float((a[1].double().cpu() * (var64 + 1e-5).sqrt() - 1).abs().max()) < 2e-5       # invstd against fp64


def test_producers_publish_their_own_maximum(h2_mode, H):
    """The elementwise kernels and the conv epilogue publish max|output| themselves (per-workgroup partial maxima);
    the consuming contraction then needs no pass over the tensor."""
    g = _gen(11)
    B, C, T = 3, 64, 200
    y = torch.randn(B, C, T, generator=g).cuda() * 3
    res = torch.randn(B, C, T, generator=g).cuda()
    scale, shift = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()

    def published(t):
        ver, ptr, slot = t._bm_amax[:3]
        assert ver == t._version and ptr == t.data_ptr()
        return float(slot.max())

    out = H.affine_act_res(y, scale, shift, res, H.ACT_GELU)
    assert published(out) == float(out.abs().max())
    u = torch.randn(B, 2 * C, T, generator=g).cuda()
    o = H.glu_fwd(u)
    assert published(o) == float(o.abs().max())
    du, _ = H.glu_bwd(o, u)
    assert published(du) == float(du.abs().max())
    assert torch.equal(H.row_amax_of(du), du.abs().amax(dim=(0, 2)))           # per channel: rows of the weight gradient
    mean, invstd = torch.randn(C, generator=g).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    dy, _, _, _ = H.act_bn_bwd(res, y, scale, shift, mean, invstd, True, H.ACT_GELU, want_affine_grads=True)
    assert published(dy) == float(dy.abs().max())
    assert torch.equal(H.row_amax_of(dy), dy.abs().amax(dim=(0, 2)))
    w = (torch.randn(128, C, 3, generator=g) / 14).cuda()
    for kw in (dict(), dict(bias=shift.new_zeros(128), scale=scale.new_ones(128), shift=shift.new_zeros(128),
                             act=H.ACT_GELU, want_pre=True)):
        _, yo, _ = H.conv_nn(out, H.pack_conv_fwd(w, (T, 2)), 128, 3, 2, **kw)
        assert published(yo) == float(yo.abs().max())
    before = H.amax_scans
    assert H.amax(yo) is yo._bm_amax[2] and H.amax_scans == before
    yo.add_(1.0)                                   # in-place change: the published value no longer applies
    H.amax(yo)
    assert H.amax_scans == before + 1


def test_training_step_needs_few_standalone_amax_passes(h2_mode, H):
    from brainmagick_amd import synthetic
    from brainmagick_amd.solver import Solver
    c = synthetic.CONFIGS["cfg2"]
    sb = synthetic.make_config_batch("cfg2", seed=3, batch=4).to("cuda")
    import test_model_gpu as TM
    solver = Solver(TM._paper_model(c["C"], c["F"], c["S"]))
    solver.train_step(sb)
    before = H.amax_scans
    solver.train_step(sb)
    scans = H.amax_scans - before
    print("stand-alone amax passes per training step:", scans)
    assert scans <= 6, scans       # the model input, the candidates and ClipLoss' operands


def test_parameters_are_packed_once_per_step(h2_mode, H):
    """All conv parameters of a model are re-packed by ONE launch per optimizer step (hip_ops._PackPlan); edits
    through torch (version counter) and through the library (weights_changed) both invalidate the packed copies."""
    from brainmagick_amd import synthetic
    from brainmagick_amd.models import SimpleConv
    from brainmagick_amd.solver import Solver
    from oracle import bm_oracle as O
    cfg = dict(O.CLIP_CONV_CFG)
    cfg.update(merger_pos_dim=32, merger_channels=24, initial_linear=24, merger_dropout=0.0)
    torch.manual_seed(0)
    # wide enough (T > 128, M >= 96) for the f16x2 conv kernel, whose packed layout the plan manages
    model = SimpleConv(in_channels={"meg": 20}, out_channels=12, hidden={"meg": 96}, n_subjects=3, **cfg)
    solver = Solver(model)
    sb = synthetic.make_batch(4, 20, 160, 12, 3, seed=5)
    solver.train_step(sb)
    n0 = H.pack_launches
    losses = [float(solver.train_step(sb)) for _ in range(3)]
    assert H.pack_launches - n0 == 3, H.pack_launches - n0
    # evaluation after the last step: one refresh (the step changed the weights), then none
    solver.eval_step(sb)
    n1 = H.pack_launches
    e1 = float(solver.eval_step(sb))
    assert H.pack_launches == n1
    # an in-place edit through torch is seen (version counter) ...
    with torch.no_grad():
        next(p for p in model.parameters() if p.dim() == 3 and p.shape[2] == 3).mul_(1.5)
    e2 = float(solver.eval_step(sb))
    assert H.pack_launches == n1 + 1 and e2 != e1
    # ... and the batched packing gives the same numbers as per-conv packing (the path non-leaf / frozen weights take:
    # one bm_pack_weights_h2 launch per conv)
    for p in model.parameters():
        p.requires_grad_(False)
    try:
        n2 = H.pack_launches
        e3 = float(solver.eval_step(sb))
        assert H.pack_launches - n2 > 1, "frozen weights should have been packed conv by conv"
    finally:
        for p in model.parameters():
            p.requires_grad_(True)
    assert e3 == e2, (e2, e3)
    assert all(math.isfinite(v) for v in losses)


def test_operand_spanning_2gb_takes_the_64bit_path(h2_mode, H):
    """The wide kernels address a segment with 32-bit byte offsets; an operand whose rows span 2 GB or more
    (2 048 wav2vec2-sized candidates on 8 GPUs) must be routed to the 64-bit-addressed kernel, not rejected."""
    g = _gen(21)
    M, Cn, T, stride = 256, 256, 512, 2_100_000                     # (Cn - 1) * stride * 4 bytes > 2 GB
    a = torch.randn(M, T, generator=g)
    rows = torch.randn(Cn, T, generator=g)
    big = torch.zeros(Cn * stride, device="cuda")
    big.view(Cn, stride)[:, :T] = rows.cuda()
    part = H.gemm_nt_partials(a.cuda(), big, 1, M, Cn, T, (0, T), (0, stride))
    ref = a.double() @ rows.double().t()
    assert rel_l2(part.sum(0), ref) < FWD_TOL


@pytest.mark.parametrize("M,Cn,T,B,G", [(270, 208, 360, 96, 7),      # composed front end: taken with the operands swapped (256 x 128 tiles)
                                        (320, 192, 192, 64, 3),      # 320 x 192 tiles as given
                                        (45, 37, 120, 9, 4)])        # too small: narrow kernels, same contract
def test_grouped_weight_gradient_on_the_wide_kernels(h2_mode, H, M, Cn, T, B, G):
    """out[g][m][c] = sum over the segments of group g and t of a[s][m][t] x[s][c][t] (the per-(layout, subject) weight
    gradient of the composed front end, bm/models/common.py:55-58 + :355-358 backward) on the wide f16x2 tiles: groups
    through order / seg (one of them EMPTY), several splits per group, arbitrary output strides."""
    g = _gen(M + Cn + G)
    a = torch.randn(B, M, T, generator=g, dtype=torch.float64)
    x = torch.randn(B, Cn, T, generator=g, dtype=torch.float64)
    idx = torch.randint(0, G - 1, (B,), generator=g)           # group G - 1 stays empty
    ref = torch.zeros(G, M, Cn, dtype=torch.float64)
    for b in range(B):
        ref[idx[b]] += a[b] @ x[b].t()
    order, seg = H.group_by_index(idx.cuda(), G)
    ag, xg = a.float().cuda(), x.float().cuda()
    labels = []
    timer = H.KernelTimer()
    H.set_kernel_timer(timer)
    try:
        out = H.gemm_nt(ag, xg, B, M, Cn, T, 1, 1, order=order, seg=seg, G=G)
        # the composed front end's layout: [G][M][Cn + 1] with the last column left alone
        aug = torch.full((G, M, Cn + 1), 7.0, device="cuda")
        H.gemm_nt(ag, xg, B, M, Cn, T, 1, 1, order=order, seg=seg, G=G, out=aug,
                  out_strides=(M * (Cn + 1), Cn + 1, 1, 0))
    finally:
        H.set_kernel_timer(None)
        labels = sorted({r[0] for r in timer.records})
    assert rel_l2(out.view(G, M, Cn), ref) < GRAD_TOL, rel_l2(out.view(G, M, Cn), ref)
    assert float(out.view(G, M, Cn)[G - 1].abs().max()) == 0.0
    assert rel_l2(aug[:, :, :Cn], ref) < GRAD_TOL and bool((aug[:, :, Cn] == 7.0).all())
    wide = bool(H.lib().bm_gemm_nt_h2_covers(M, Cn, 1, B, T, G, 1, 1)) or \
        bool(H.lib().bm_gemm_nt_h2_covers(Cn, M, 1, B, T, G, 1, 1))
    assert wide == (M >= 200), (M, wide)
    assert all(("_h2w" in lb) == wide for lb in labels), labels
