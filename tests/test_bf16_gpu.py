"""OPT-IN bf16-operand MFMA compute mode (brainmagick_amd.set_compute_dtype("bf16")): fp32
accumulate, fp32 activations in HBM.  Tolerances per SURVEY.md §8d for bf16 inputs: per-op forward
rel-L2 <= 1e-2 (observed ~2-3e-3), loss |delta| <= 2e-2.  The fp32 mode stays the parity-green default."""
import math

import pytest
import torch
from torch.nn import functional as F

from helpers import rel_l2
from oracle import bm_oracle as O

pytestmark = pytest.mark.gpu
BF16_TOL = 1e-2


@pytest.fixture()
def bf16_mode():
    import brainmagick_amd
    default = brainmagick_amd.get_compute_dtype()
    brainmagick_amd.set_compute_dtype("bf16")
    yield
    brainmagick_amd.set_compute_dtype(default)


@pytest.mark.parametrize("Cin,M,KS,dil,T,B", [(270, 320, 3, 1, 360, 3), (320, 320, 3, 16, 343, 2),
                                               (320, 640, 3, 1, 361, 2), (640, 120, 1, 1, 97, 3),
                                               (33, 40, 5, 8, 100, 2), (20, 12, 1, 1, 48, 5),
                                               (5, 1024, 1, 1, 130, 1)])
def test_conv_nn_bf16(bf16_mode, Cin, M, KS, dil, T, B):
    from brainmagick_amd import hip_ops as H
    g = torch.Generator().manual_seed(Cin + M + KS + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(M, Cin, KS, generator=g) / math.sqrt(Cin * KS)
    b = torch.randn(M, generator=g)
    res = torch.randn(B, M, T, generator=g)
    ref = F.conv1d(x.double(), w.double(), b.double(), padding=KS // 2 * dil, dilation=dil)
    wp = H.pack_conv_fwd(w.cuda())
    assert wp.dtype == torch.bfloat16
    pre, out, _ = H.conv_nn(x.cuda(), wp, M, KS, dil, bias=b.cuda(), res=res.cuda(), act=H.ACT_GELU,
                            want_pre=True)
    err = rel_l2(pre, ref)
    assert 1e-5 < err < BF16_TOL, err          # really bf16 operands, and within tolerance
    assert rel_l2(out, F.gelu(ref) + res.double()) < BF16_TOL
    # bf16-rounded operands reproduce the kernel to fp32 accuracy (only the rounding differs)
    xr, wr = x.bfloat16().double(), w.bfloat16().double()
    ref_r = F.conv1d(xr, wr, b.double(), padding=KS // 2 * dil, dilation=dil)
    assert rel_l2(pre, ref_r) < 2e-5
    # data gradient through the same kernel
    dy = torch.randn(B, M, T, generator=g)
    xg = x.double().requires_grad_(True)
    F.conv1d(xg, w.double(), None, padding=KS // 2 * dil, dilation=dil).backward(dy.double())
    _, dx, _ = H.conv_nn(dy.cuda(), H.pack_conv_dgrad(w.cuda()), Cin, KS, dil)
    assert rel_l2(dx, xg.grad) < BF16_TOL


@pytest.mark.parametrize("Cin,M,KS,dil,T,B", [(320, 320, 3, 1, 360, 3), (320, 320, 3, 2, 343, 2),
                                               (320, 640, 3, 16, 360, 2), (270, 270, 1, 1, 361, 2),
                                               (33, 40, 5, 8, 100, 2), (640, 120, 1, 1, 97, 3)])
def test_gemm_nt_bf16_weight_grad(bf16_mode, Cin, M, KS, dil, T, B):
    from brainmagick_amd import hip_ops as H
    g = torch.Generator().manual_seed(Cin * 3 + M + KS + dil + T)
    x = torch.randn(B, Cin, T, generator=g)
    dy = torch.randn(B, M, T, generator=g)
    w = torch.zeros(M, Cin, KS, dtype=torch.float64, requires_grad=True)
    F.conv1d(x.double(), w, None, padding=KS // 2 * dil, dilation=dil).backward(dy.double())
    dw = H.gemm_nt(dy.cuda(), x.cuda(), B, M, Cin, T, KS, dil)[0]
    err = rel_l2(dw, w.grad)
    assert 1e-5 < err < BF16_TOL, err
    w2 = torch.zeros(M, Cin, KS, dtype=torch.float64, requires_grad=True)
    F.conv1d(x.bfloat16().double(), w2, None, padding=KS // 2 * dil, dilation=dil) \
        .backward(dy.bfloat16().double())
    assert rel_l2(dw, w2.grad) < 2e-5          # only the operand rounding differs
    # grouped (per-subject) form
    if KS == 1 and Cin == 270:
        subj = torch.tensor([1, 0])
        order, seg = H.group_by_index(subj.cuda(), 2)
        out = torch.empty(2, Cin, M, device="cuda")
        H.gemm_nt(dy.cuda(), x.cuda(), B, M, Cin, T, 1, 1, order=order, seg=seg, G=2, out=out,
                  out_strides=(Cin * M, 1, M, 0))
        ref = torch.einsum("bct,bdt->bcd", x.double(), dy.double())[[1, 0]]
        assert rel_l2(out, ref) < BF16_TOL


def test_bf16_training_step_against_fp32_oracle(bf16_mode):
    """Whole SimpleConv + ClipLoss step in bf16 compute mode vs the fp32 CPU oracle: loss within 2e-2,
    estimate within 1e-2 rel-L2, gradients within 5e-2 rel-L2 (bf16 operands in ~45 chained GEMMs);
    ClipLoss scores and the attention logits stay fp32."""
    import copy
    from brainmagick_amd import synthetic
    from brainmagick_amd.models import SimpleConv
    from brainmagick_amd.solver import Solver
    c = synthetic.CONFIGS["cfg2"]
    B, T, Fd = 8, 360, c["F"]
    sb = synthetic.make_batch(B, c["C"], T, Fd, c["S"], seed=2036)
    torch.manual_seed(0)
    model = SimpleConv(in_channels={"meg": c["C"]}, out_channels=Fd, hidden={"meg": 320},
                       n_subjects=c["S"], **O.CLIP_CONV_CFG)
    oracle = O.OracleModel(copy.deepcopy(model.state_dict()), O.CLIP_CONV_CFG, 320, Fd)
    ban = torch.tensor([0.3, 0.7])
    model.merger.ban_center_override = ban
    solver = Solver(model)
    loss_ref, est_ref, grads_ref = oracle.loss_and_grads(sb.meg, sb.positions(), sb.subject_index,
                                                         sb.features, True, ban)
    loss = solver.train_step(sb)
    assert abs(float(loss) - float(loss_ref)) < 2e-2, (float(loss), float(loss_ref))
    gscale = max(float(v.norm()) for v in grads_ref.values())
    worst = 0.0
    for k, p in model.named_parameters():
        if float(grads_ref[k].abs().max()) <= 1e-5 * gscale:
            continue
        worst = max(worst, rel_l2(p.grad, grads_ref[k]))
    assert worst < 5e-2, worst
