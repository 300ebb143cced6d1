"""The fp32-class compute modes side by side.

The library default is "f16x2" (fp32-accurate contractions on the f16 matrix cores: two scaled f16 planes per
operand, three partial products, fp32 accumulate); every other GPU test file runs in it.  This file re-runs the
kernel, golden-vector and oracle suites in the OTHER two fp32-class modes -- "f32" (exact-fp32 MFMA,
v_mfma_f32_32x32x2_f32, bit-comparable to an fp32 FMA chain) and "f32x3" (exact 3-way bf16 split, six partial
products) -- at the SAME tolerances: per-kernel forward rel-L2 <= 5e-6 vs fp64, gradients <= 2e-5, end-to-end
1e-5 / 1e-4 / loss 1e-4 -- and compares the modes' errors directly."""
import math

import pytest
import torch
from torch.nn import functional as F

import test_kernels_gpu as TK
import test_model_gpu as TM
from helpers import rel_l2, MODEL_FIXTURES

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["f32", "f32x3"])
def x3_mode(request):
    """(historical name) switches to another fp32-class mode for the test, then back to the default."""
    import brainmagick_amd
    import os
    default = brainmagick_amd.get_compute_dtype()
    if "BM_COMPUTE_DTYPE" not in os.environ:
        assert default == "f16x2"          # the library default (an explicit env override is respected)
    brainmagick_amd.set_compute_dtype(request.param)
    yield
    brainmagick_amd.set_compute_dtype(default)


@pytest.fixture(scope="module")
def H():
    from brainmagick_amd import hip_ops
    return hip_ops


@pytest.mark.parametrize("Cin,M,KS,dil,T,B", TK.CONV_CASES)
def test_conv_forward_x3(x3_mode, H, Cin, M, KS, dil, T, B):
    TK.test_conv_nn_forward(H, Cin, M, KS, dil, T, B)


@pytest.mark.parametrize("Cin,M,KS,dil,T,B", TK.CONV_CASES[:7])
def test_conv_backward_x3(x3_mode, H, Cin, M, KS, dil, T, B):
    TK.test_conv_backward_kernels(H, Cin, M, KS, dil, T, B)


def test_x3_error_is_fp32_class(H):
    """Error vs fp64 of the x3 path next to the exact-fp32 MFMA path on the same inputs, and proof that
    the three bf16 planes are really used (a single-plane bf16 result would be ~2e-3 off)."""
    import brainmagick_amd
    g = torch.Generator().manual_seed(0)
    B, Cin, M, KS, dil, T = 4, 320, 320, 3, 2, 360
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(M, Cin, KS, generator=g) / math.sqrt(Cin * KS)
    dy = torch.randn(B, M, T, generator=g)
    ref = F.conv1d(x.double(), w.double(), None, padding=dil, dilation=dil)
    wg = torch.zeros(M, Cin, KS, dtype=torch.float64, requires_grad=True)
    F.conv1d(x.double(), wg, None, padding=dil, dilation=dil).backward(dy.double())
    errs = {}
    default = brainmagick_amd.get_compute_dtype()
    for mode in ("f32", "f32x3"):
        brainmagick_amd.set_compute_dtype(mode)
        try:
            _, y, _ = H.conv_nn(x.cuda(), H.pack_conv_fwd(w.cuda()), M, KS, dil)
            dw = H.gemm_nt(dy.cuda(), x.cuda(), B, M, Cin, T, KS, dil)[0]
        finally:
            brainmagick_amd.set_compute_dtype(default)
        errs[mode] = (rel_l2(y, ref), rel_l2(dw, wg.grad))
    print("rel-L2 vs fp64 (conv fwd, wgrad):", errs)
    for k in range(2):
        assert errs["f32x3"][k] < 1e-6
        assert errs["f32x3"][k] < 3 * errs["f32"][k] + 1e-7


def test_subject_layers_and_merger_x3(x3_mode, H):
    TK.test_subject_layers_kernels(H)
    TK.test_merger_kernels(H)


@pytest.mark.parametrize("B,Bc,Fd,T", [(6, 6, 10, 48), (5, 12, 7, 33), (64, 64, 120, 360)])
def test_clip_x3(x3_mode, H, B, Bc, Fd, T):
    TK.test_clip_kernels(H, B, Bc, Fd, T)


@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_reference_golden_x3(x3_mode, name):
    TM.test_against_reference_golden(name)


@pytest.mark.parametrize("cfg_name,B,T", [("cfg2", 8, 360), ("cfg5", 6, 343)])
def test_paper_model_step_x3(x3_mode, cfg_name, B, T):
    TM.test_paper_model_step_against_oracle(cfg_name, B, T)


def test_training_curve_x3(x3_mode):
    # the 200-step run is the default (f16x2) mode's; the cross-check modes follow the oracle for 40 steps
    TM.test_training_curve_and_top10_parity(steps=40, n_held=512)
