"""Per-kernel parity: each libbmhip entry point (called through the C-ABI) vs the CPU oracle /
an fp64 torch CPU restatement of the same op, on seeded inputs.  fp32 tolerances are written in
each test (rel-L2 unless stated)."""
import math

import pytest
from pathlib import Path
import torch
from torch.nn import functional as F

from helpers import rel_l2
from oracle import bm_oracle as O

pytestmark = pytest.mark.gpu

FWD_TOL = 5e-6     # fp32 MFMA (exact fp32 FMA chain) vs fp64 reference
GRAD_TOL = 2e-5


@pytest.fixture(scope="module")
def H():
    from brainmagick_amd import hip_ops
    return hip_ops


def _gen(seed):
    return torch.Generator().manual_seed(seed)


CONV_CASES = [
    # Cin, M, KS, dil, T, B
    (270, 320, 3, 1, 360, 3),
    (320, 320, 3, 16, 343, 2),
    (320, 640, 3, 1, 361, 2),
    (320, 640, 1, 1, 360, 2),
    (640, 120, 1, 1, 97, 3),
    (20, 12, 1, 1, 48, 5),
    (33, 40, 5, 8, 100, 2),
    (16, 16, 3, 4, 7, 1),      # T smaller than the dilation halo
    (5, 1024, 1, 1, 130, 1),
    # shapes of the wide 320-row tiles (M % 320 == 0, Cin % 64 == 0)
    (320, 320, 3, 2, 360, 4),
    (64, 640, 3, 16, 343, 3),
    (128, 320, 3, 1, 361, 3),
    (40, 320, 3, 7, 577, 1),       # Cin not a multiple of 16, odd dilation, last 192-column tile holds 1 column
    (16, 320, 1, 1, 129, 1),       # a single, mostly empty time tile
]


@pytest.mark.parametrize("Cin,M,KS,dil,T,B", CONV_CASES)
def test_conv_nn_forward(H, Cin, M, KS, dil, T, B):
    g = _gen(Cin * 7 + M + KS + dil + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(M, Cin, KS, generator=g) / math.sqrt(Cin * KS)
    b = torch.randn(M, generator=g)
    ref = F.conv1d(x.double(), w.double(), b.double(), padding=KS // 2 * dil, dilation=dil)
    wp = H.pack_conv_fwd(w.cuda(), (T, dil))
    _, y, _ = H.conv_nn(x.cuda(), wp, M, KS, dil, bias=b.cuda())
    assert y.shape == ref.shape
    assert rel_l2(y, ref) < FWD_TOL


def test_conv_nn_epilogue_and_stats(H):
    g = _gen(5)
    B, Cin, M, T, KS, dil = 3, 48, 70, 200, 3, 2
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(M, Cin, KS, generator=g) / math.sqrt(Cin * KS)
    b = torch.randn(M, generator=g)
    scale = torch.rand(M, generator=g) + 0.5
    shift = torch.randn(M, generator=g)
    res = torch.randn(B, M, T, generator=g)
    pre_ref = F.conv1d(x.double(), w.double(), b.double(), padding=dil, dilation=dil)
    for act, fn in [(H.ACT_GELU, F.gelu), (H.ACT_RELU, F.relu),
                    (H.ACT_LEAKY, lambda z: F.leaky_relu(z, 0.1)), (H.ACT_NONE, lambda z: z)]:
        out_ref = fn(pre_ref * scale.double()[None, :, None] + shift.double()[None, :, None]) \
            + res.double()
        pre, out, stats = H.conv_nn(x.cuda(), H.pack_conv_fwd(w.cuda(), (x.shape[2], dil)), M, KS, dil, bias=b.cuda(),
                                    scale=scale.cuda(), shift=shift.cuda(), res=res.cuda(), act=act,
                                    leak=0.1, want_pre=True, want_stats=True)
        assert rel_l2(pre, pre_ref) < FWD_TOL
        assert rel_l2(out, out_ref) < FWD_TOL
        s = stats.double().cpu().sum(0)
        assert rel_l2(s[:, 0], pre_ref.sum((0, 2))) < 1e-5
        assert rel_l2(s[:, 1], (pre_ref ** 2).sum((0, 2))) < 1e-5


@pytest.mark.parametrize("Cin,M,KS,dil,T,B", CONV_CASES[:7] + CONV_CASES[9:])
def test_conv_backward_kernels(H, Cin, M, KS, dil, T, B):
    g = _gen(Cin + M * 3 + KS + dil + T)
    x = torch.randn(B, Cin, T, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(M, Cin, KS, generator=g, dtype=torch.float64) / math.sqrt(Cin * KS)) \
        .requires_grad_(True)
    dy = torch.randn(B, M, T, generator=g, dtype=torch.float64)
    y = F.conv1d(x, w, None, padding=KS // 2 * dil, dilation=dil)
    y.backward(dy)
    dyg, xg, wg = dy.float().cuda(), x.detach().float().cuda(), w.detach().float().cuda()
    _, dx, _ = H.conv_nn(dyg, H.pack_conv_dgrad(wg, (T, dil)), Cin, KS, dil)
    assert rel_l2(dx, x.grad) < GRAD_TOL
    dw = H.gemm_nt(dyg, xg, B, M, Cin, T, KS, dil)
    assert dw.shape == (1, M, Cin, KS)
    assert rel_l2(dw[0], w.grad) < GRAD_TOL
    for nsplit in (1, 3, 4, 50):     # 50: one or two 16-sample stages per workgroup in the wide-tile kernel
        dw2 = H.gemm_nt(dyg, xg, B, M, Cin, T, KS, dil, nsplit=nsplit)
        assert rel_l2(dw2[0], w.grad) < GRAD_TOL


def test_conv_deterministic(H):
    g = _gen(11)
    x = torch.randn(4, 64, 360, generator=g).cuda()
    dy = torch.randn(4, 96, 360, generator=g).cuda()
    a = H.gemm_nt(dy, x, 4, 96, 64, 360, 3, 2)
    b = H.gemm_nt(dy, x, 4, 96, 64, 360, 3, 2)
    assert torch.equal(a, b)


def test_subject_layers_kernels(H):
    """bm/models/common.py:55-58 forward and both gradients, grouped by subject."""
    g = _gen(3)
    B, C, D, T, S = 9, 37, 45, 120, 4
    x = torch.randn(B, C, T, generator=g, dtype=torch.float64, requires_grad=True)
    W = torch.randn(S, C, D, generator=g, dtype=torch.float64, requires_grad=True)
    subj = torch.tensor([2, 0, 2, 3, 3, 3, 0, 2, 2])        # subject 1 is absent from the batch
    ref = O.subject_layers(x, W, subj)
    dy = torch.randn(B, D, T, generator=g, dtype=torch.float64)
    ref.backward(dy)
    xg, Wg, dyg = x.detach().float().cuda(), W.detach().float().cuda(), dy.float().cuda()
    widx = subj.to(torch.int32).cuda()
    wp = H.pack_weights(Wg, S, D, C, 1, C * D, 1, D, 0, shape=(T, 1))
    _, y, _ = H.conv_nn(xg, wp, D, 1, 1, widx=widx)
    assert rel_l2(y, ref) < FWD_TOL
    wpt = H.pack_weights(Wg, S, C, D, 1, C * D, D, 1, 0, shape=(T, 1))
    _, dx, _ = H.conv_nn(dyg, wpt, C, 1, 1, widx=widx)
    assert rel_l2(dx, x.grad) < GRAD_TOL
    order, seg = H.group_by_index(subj.cuda(), S)
    assert seg.tolist() == [0, 2, 2, 6, 9]
    assert order.tolist() == [1, 6, 0, 2, 7, 8, 3, 4, 5]
    dW = torch.empty(S, C, D, device="cuda")
    H.gemm_nt(dyg, xg, B, D, C, T, 1, 1, order=order, seg=seg, G=S, out=dW,
              out_strides=(C * D, 1, D, 0))
    assert rel_l2(dW, W.grad) < GRAD_TOL
    assert torch.count_nonzero(dW[1]) == 0


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("B,C,T", [(256, 320, 360),      # the production shape: 10 splits of <= 26 segments per channel
                                   (173, 320, 360),      # ragged batch: splits of 17 / 18 segments
                                   (37, 64, 192),        # one slab shorter than a workgroup's first trip
                                   (6, 40, 360), (1, 8, 4), (64, 1500, 48)])
def test_one_pass_batchnorm_backward_against_the_two_pass_kernels(H, B, C, T, train):
    """bm_act_bn_bwd in train mode runs as ONE pass when T % 4 == 0 and a split's slab fits the registers
    (bn_bwd_fused_kernel: the workgroups of a channel exchange their partial sums inside the launch): same results as
    the two-pass kernels (to summation order), the fp64 formula, and bit-identical from run to run.  ``train=False``:
    eval-mode BatchNorm with affine gradients takes the same kernel (the sums only feed dgamma / dbeta)."""
    from brainmagick_amd._lib import lib
    g = _gen(B + C + T)
    y = torch.randn(B, C, T, generator=g) * 1.5 + 0.3
    dout = torch.randn(B, C, T, generator=g)
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g)
    yd = y.double()
    mean = yd.mean((0, 2))
    invstd = 1.0 / torch.sqrt(yd.var((0, 2), unbiased=False) + 1e-5)
    scale = gamma.double() * invstd
    shift = beta.double() - mean * scale
    args = [t.float().cuda() for t in (dout, y, scale, shift, mean, invstd)]

    def run(mode):
        prev = lib().bm_act_bn_bwd_set_fused(mode)
        try:
            if mode:
                assert lib().bm_act_bn_bwd_fused_covers(B, C, T) == 1
            dy, dgamma, dbeta, dbias = H.act_bn_bwd(*args, train, H.ACT_GELU, want_affine_grads=True)
            amax, rows = H.amax(dy).clone(), H.row_amax_of(dy)
            torch.cuda.synchronize()
            return dy, dgamma, dbeta, dbias, amax, (rows.clone() if rows is not None else None)
        finally:
            lib().bm_act_bn_bwd_set_fused(prev)

    two = run(0)
    # fp64 formula
    z = yd * scale[None, :, None] + shift[None, :, None]
    dz = dout.double() * (0.5 * (1 + torch.erf(z / math.sqrt(2))) + z * torch.exp(-0.5 * z * z) / math.sqrt(2 * math.pi))
    xh = (yd - mean[None, :, None]) * invstd[None, :, None]
    n = B * T
    ref = scale[None, :, None] * (dz - dz.sum((0, 2), keepdim=True) / n - xh * (dz * xh).sum((0, 2), keepdim=True) / n)
    if not train:
        ref = scale[None, :, None] * dz
    fallbacks0 = lib().bm_act_bn_bwd_fused_fallbacks()
    results = [("two-pass", two)]
    for mode in (1,):
        one, again = run(mode), run(mode)
        for a, b in zip(one, again):
            assert a is None or torch.equal(a, b)                     # deterministic
        results.append((f"one-pass mode {mode}", one))
        # partners that never publish in time: every workgroup computes the other splits' sums itself -- same bits
        assert lib().bm_act_bn_bwd_fused_fallbacks() == fallbacks0, "a workgroup gave up on its partners on an idle chip"
        limit = lib().bm_act_bn_bwd_set_poll_limit(0)
        try:
            alone = run(mode)
        finally:
            lib().bm_act_bn_bwd_set_poll_limit(limit)
        if lib().bm_act_bn_bwd_fused_covers(B, C, T) and B >= 8:          # (B < 8: one split per channel, no partner)
            assert lib().bm_act_bn_bwd_fused_fallbacks() > fallbacks0      # the forced fallback is counted
        for a, b in zip(one, alone):
            assert a is None or torch.equal(a, b), f"mode {mode}: the fallback path rounds differently"
        if B * C * T >= 1 << 24:
            # the hand-off under UNEVEN load (cdna_hip_programming.md, Guideline 16: "test every hand-off under uneven
            # load"): a second stream streams 512 MB per launch through the chip while the kernel runs, so the
            # workgroups of a channel start and finish at different times -- same bits
            side = torch.cuda.Stream()
            noise = torch.zeros(64 << 20, device="cuda")
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(24):
                    noise.add_(1.0)
            loaded = run(mode)
            side.synchronize()
            for a, b in zip(one, loaded):
                assert a is None or torch.equal(a, b), f"mode {mode}: different bits under concurrent load"
        assert rel_l2(one[0], two[0]) < 2e-6
        assert float(one[4].max()) == float(one[0].abs().max())      # the published maximum is that of what was written
        if one[5] is not None:
            assert torch.equal(one[5].cpu(), one[0].abs().amax((0, 2)).cpu())
        if train:
            assert float(one[3].abs().max()) < 1e-2 * max(1.0, float(one[0].abs().max()))     # sum(dy) of a BN input is ~0
    assert len(results) > 1
    for name, got in results:
        assert rel_l2(got[0], ref) < GRAD_TOL, name
        assert rel_l2(got[1], (dz * xh).sum((0, 2))) < GRAD_TOL and rel_l2(got[2], dz.sum((0, 2))) < GRAD_TOL, name


@pytest.mark.parametrize("T", [360, 343])
@pytest.mark.parametrize("act", ["gelu", "relu", "leaky"])
def test_batchnorm_act_residual(H, T, act):
    """conv -> BatchNorm1d(train) -> act -> + residual, forward and backward (common.py:113-151)."""
    g = _gen(T)
    B, C = 6, 40
    code = {"gelu": H.ACT_GELU, "relu": H.ACT_RELU, "leaky": H.ACT_LEAKY}[act]
    fn = {"gelu": F.gelu, "relu": F.relu, "leaky": lambda z: F.leaky_relu(z, 0.1)}[act]
    y = (torch.randn(B, C, T, generator=g, dtype=torch.float64) * 2 + 0.7).requires_grad_(True)
    res = torch.randn(B, C, T, generator=g, dtype=torch.float64)
    gamma = (torch.rand(C, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    beta = torch.randn(C, generator=g, dtype=torch.float64).requires_grad_(True)
    rm = torch.randn(C, generator=g, dtype=torch.float64)
    rv = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    out_ref = fn(F.batch_norm(y, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.1,
                              eps=1e-5)) + res
    dout = torch.randn(B, C, T, generator=g, dtype=torch.float64)
    out_ref.backward(dout)

    # per-tile (sum, sumsq) partials as conv_nn's epilogue would produce them
    yf = y.detach().float()
    ntn = (T + 127) // 128
    stats = torch.zeros(B * ntn, C, 2)
    for b in range(B):
        for n in range(ntn):
            sl = yf[b, :, n * 128:(n + 1) * 128]
            stats[b * ntn + n, :, 0] = sl.sum(1)
            stats[b * ntn + n, :, 1] = (sl * sl).sum(1)
    rm_g, rv_g = rm.float().cuda(), rv.float().cuda()
    nb = torch.zeros((), dtype=torch.int64, device="cuda")
    mean, invstd, scale, shift = H.bn_finalize(stats.cuda(), B * T, gamma.detach().float().cuda(),
                                               beta.detach().float().cuda(), rm_g, rv_g, nb, 0.1, 1e-5)
    assert int(nb) == 1
    assert rel_l2(rm_g, rm_ref) < 1e-6 and rel_l2(rv_g, rv_ref) < 1e-6
    out = H.affine_act_res(yf.cuda(), scale, shift, res.float().cuda(), code, 0.1)
    assert rel_l2(out, out_ref) < FWD_TOL
    dy, dgamma, dbeta, dbias = H.act_bn_bwd(dout.float().cuda(), yf.cuda(), scale, shift, mean, invstd,
                                            True, code, 0.1, want_affine_grads=True)
    assert rel_l2(dy, y.grad) < GRAD_TOL
    assert rel_l2(dgamma, gamma.grad) < GRAD_TOL
    assert rel_l2(dbeta, beta.grad) < GRAD_TOL
    assert dbias.abs().max().item() < 1e-3          # sum(dy) of a BN input is ~0

    # eval mode: running statistics
    y2 = y.detach().clone().requires_grad_(True)
    g2 = gamma.detach().clone().requires_grad_(True)
    b2 = beta.detach().clone().requires_grad_(True)
    out_ref = fn(F.batch_norm(y2, rm, rv, g2, b2, training=False, eps=1e-5))
    out_ref.backward(dout)
    mean, invstd, scale, shift = H.bn_eval_affine(gamma.detach().float().cuda(),
                                                  beta.detach().float().cuda(), rm.float().cuda(),
                                                  rv.float().cuda(), 1e-5)
    out = H.affine_act_res(yf.cuda(), scale, shift, None, code, 0.1)
    assert rel_l2(out, out_ref) < FWD_TOL
    dy, dg_e, db_e, dbias = H.act_bn_bwd(dout.float().cuda(), yf.cuda(), scale, shift, mean, invstd,
                                         False, code, 0.1, want_affine_grads=True)
    assert rel_l2(dy, y2.grad) < GRAD_TOL
    assert rel_l2(dbias, y2.grad.sum((0, 2))) < GRAD_TOL
    assert rel_l2(dg_e, g2.grad) < GRAD_TOL and rel_l2(db_e, b2.grad) < GRAD_TOL
    # no BN at all
    y3 = y.detach().clone().requires_grad_(True)
    fn(y3).backward(dout)
    dy, _, _, _ = H.act_bn_bwd(dout.float().cuda(), yf.cuda(), None, None, None, None, False, code, 0.1)
    assert rel_l2(dy, y3.grad) < GRAD_TOL


@pytest.mark.parametrize("T", [360, 79])
def test_glu_and_channel_sum(H, T):
    g = _gen(T + 1)
    B, Hc = 5, 24
    u = torch.randn(B, 2 * Hc, T, generator=g, dtype=torch.float64, requires_grad=True)
    ref = F.glu(u, dim=1)
    dout = torch.randn(B, Hc, T, generator=g, dtype=torch.float64)
    ref.backward(dout)
    out = H.glu_fwd(u.detach().float().cuda())
    assert rel_l2(out, ref) < FWD_TOL
    du, dbias = H.glu_bwd(dout.float().cuda(), u.detach().float().cuda())
    assert rel_l2(du, u.grad) < GRAD_TOL
    assert rel_l2(dbias, u.grad.sum((0, 2))) < GRAD_TOL
    assert rel_l2(H.channel_sum(du), u.grad.sum((0, 2))) < GRAD_TOL


def test_merger_kernels(H):
    """FourierEmb + masked softmax over sensors + weighted reduction (common.py:239-271,334-358)."""
    g = _gen(8)
    U, C, Oc, D, T = 3, 45, 30, 128, 90
    pos = torch.rand(U, C, 2, generator=g)
    pos[1, 30:] = O.INVALID
    heads = torch.randn(Oc, D, generator=g) / D ** 0.5
    emb_ref = O.fourier_emb(pos, D)
    emb = H.fourier_emb(pos.cuda(), D)
    assert (emb.cpu() - emb_ref).abs().max().item() < 5e-6
    ban = torch.tensor([0.4, 0.6])
    for training, radius in [(False, 0.0), (True, 0.2)]:
        w_ref = O.merger_weights(heads.double(), pos.double(), training, 0.2, ban.double())
        scores = H.gemm_nt(heads.cuda(), emb, U, Oc, C, D, a_strides=(0, D), x_strides=(C * D, D),
                           seg=torch.arange(U + 1, dtype=torch.int32).cuda(), G=U)[..., 0]
        w = H.masked_softmax(scores.contiguous(), pos.cuda(), ban.cuda() if radius else None, radius)
        assert rel_l2(w, w_ref) < 2e-5
        assert torch.count_nonzero(w[1, :, 30:]) == 0
    # backward of the softmax
    wd = w_ref.clone().requires_grad_(True)
    sc = torch.randn(U, Oc, C, generator=g, dtype=torch.float64, requires_grad=True)
    sm = torch.softmax(sc, 2)
    dw = torch.randn(U, Oc, C, generator=g, dtype=torch.float64)
    sm.backward(dw)
    ds = H.softmax_bwd(sm.detach().float().cuda(), dw.float().cuda())
    assert rel_l2(ds, sc.grad) < GRAD_TOL
    del wd


@pytest.mark.parametrize("B,Bc,Fd,T", [(6, 6, 10, 48), (5, 12, 7, 33), (64, 64, 120, 360)])
def test_clip_kernels(H, B, Bc, Fd, T):
    g = _gen(B + Bc)
    est = torch.randn(B, Fd, T, generator=g, dtype=torch.float64, requires_grad=True)
    cand = torch.randn(Bc, Fd, T, generator=g, dtype=torch.float64) * 2 + 0.3
    scores_ref = O.clip_scores(est, cand)
    loss_ref = O.clip_loss(est, cand)
    loss_ref.backward()
    K = Fd * T
    eg, cg = est.detach().float().cuda(), cand.float().cuda()
    inv = H.clip_inv_norms(cg)
    assert rel_l2(inv, 1 / (1e-8 + cand.norm(dim=(1, 2)))) < 1e-6
    part = H.gemm_nt_partials(eg, cg, 1, B, Bc, K, (0, K), (0, K))
    scores, probs, dscaled, loss = H.clip_ce(part, inv, True, True, True)
    assert rel_l2(scores, scores_ref) < FWD_TOL
    assert rel_l2(probs, torch.softmax(scores_ref, 1)) < 1e-5
    assert abs(float(loss) - float(loss_ref)) < 1e-5
    wp = H.pack_weights(dscaled, 1, B, Bc, 1, 0, Bc, 1, 0, shape=(K, 1))
    _, dest, _ = H.conv_nn(cg.view(1, Bc, K), wp, B, 1, 1)
    assert rel_l2(dest.view(B, Fd, T), est.grad) < GRAD_TOL


def test_clip_loss_at_the_whole_node_shape(H):
    """cfg4 (BASELINE configs[3]): 256 estimates of one rank against the 2 048 candidates of 8 ranks (mel features,
    K = 120 * 360 -- the shape the driver's 8-GPU run puts through ClipLoss), targets = the rank's own block at
    target_offset = 7 * 256 (rank 7), a few padding candidates masked (per-rank rejection).  Loss, probabilities of the
    target block and dEst against the fp32 CPU oracle (own block rolled to the front, masked candidates removed)."""
    from brainmagick_amd.losses import ClipLoss
    B, Bc, Fd, T, off = 256, 2048, 120, 360, 7 * 256
    g = _gen(77)
    est = (torch.randn(B, Fd, T, generator=g) * 0.5).requires_grad_(True)
    cand = torch.randn(Bc, Fd, T, generator=g)
    est.data += 0.02 * cand[off:off + B]                      # planted: target scores ~4 above the rest, far from saturation
    valid = torch.ones(Bc)
    masked = [3, 300, 301, 1500, 1791]        # (none inside the target block 1792 .. 2047)
    valid[masked] = 0
    kept = torch.tensor([i for i in range(Bc) if i not in masked and not off <= i < off + B])
    cand_ref = torch.cat([cand[off:off + B], cand[kept]])     # bm/losses.py:105-111: the first B candidates are the targets
    prev = torch.get_num_threads()
    torch.set_num_threads(min(32, prev))
    try:
        loss_ref = O.clip_loss(est, cand_ref)
        loss_ref.backward()
    finally:
        torch.set_num_threads(prev)
    eg = est.detach().cuda().requires_grad_(True)
    cg = cand.cuda()
    loss = ClipLoss().cuda()(eg, cg, torch.ones(B, 1, T, dtype=torch.bool, device="cuda"), target_offset=off,
                              candidate_valid=valid.cuda())
    loss.backward()
    assert abs(float(loss) - float(loss_ref)) < 1e-4, (float(loss), float(loss_ref))
    assert rel_l2(eg.grad, est.grad) < GRAD_TOL, rel_l2(eg.grad, est.grad)
    # a masked candidate has probability exactly 0 and takes no gradient share
    from brainmagick_amd import functional as BF
    inv = H.clip_inv_norms(cg)
    part = BF._clip_raw_scores(eg.detach(), cg, B, Bc, Fd * T)
    _, probs, dscaled, _ = H.clip_ce(part, inv, want_probs=True, want_grad=True, want_loss=True, target_offset=off,
                                     col_valid=valid.cuda())
    assert float(probs[:, masked].abs().max()) == 0.0 and float(dscaled[:, masked].abs().max()) == 0.0
    assert float((probs.sum(1) - 1).abs().max()) < 1e-5


def test_adam_kernel(H):
    g = _gen(2)
    n = 10007
    p = torch.randn(n, generator=g)
    m = torch.zeros(n)
    v = torch.zeros(n)
    pg, mg, vg = p.cuda(), m.cuda(), v.cuda()
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * 0.1
        O.adam_step(p, grad, m, v, step)
        H.adam_step(pg, grad.cuda(), mg, vg, step, 3e-4, 0.9, 0.999, 1e-8)
    assert rel_l2(pg, p) < 1e-6
    assert rel_l2(mg, m) < 1e-6 and rel_l2(vg, v) < 1e-6


def test_cpu_tensors_are_rejected(H):
    from brainmagick_amd._lib import BmHipError
    with pytest.raises(BmHipError):
        H.glu_fwd(torch.randn(2, 4, 8))


def test_topk_rows_and_retrieval_accuracy(H):
    """bm_topk_rows vs torch.topk + the reference rule of scripts/run_eval_probs.py:237-264."""
    from brainmagick_amd import retrieval
    from brainmagick_amd.losses import ClipLoss
    g = _gen(21)
    probs = torch.softmax(torch.randn(37, 301, generator=g) * 3, 1)
    labels = torch.randint(0, 40, (301,), generator=g)
    rows = labels[:37].clone()
    idx, val, hits = H.topk_rows(probs.cuda(), 10, labels.cuda(), rows.cuda())
    ref = probs.topk(10, dim=1)
    assert torch.equal(idx.cpu().long(), ref.indices)
    assert torch.equal(val.cpu(), ref.values)
    for k in (1, 5, 10):
        assert retrieval.get_accuracy_from_probs(probs.cuda(), rows, labels, k) == \
            pytest.approx(O.topk_accuracy(probs, labels, rows, k))
    # fewer columns than k, ties
    small = torch.tensor([[0.5, 0.5, 0.0], [0.1, 0.2, 0.7]])
    idx, _, _ = H.topk_rows(small.cuda(), 5)
    assert idx.cpu().tolist() == [[0, 1, 2, -1, -1], [2, 1, 0, -1, -1]]
    # end to end: probabilities of every candidate, batched, vs the oracle
    est = torch.randn(50, 6, 40, generator=g)
    cand = torch.randn(70, 6, 40, generator=g)
    cand[:50] += 0.5 * est
    pr = retrieval.builds_probs(ClipLoss().cuda(), est, cand, batch_size=16)
    assert rel_l2(pr, O.clip_probabilities(est.double(), cand.double())) < 1e-5
    acc = retrieval.segment_topk_accuracy(ClipLoss().cuda(), est, cand, batch_size=16)
    ref_acc = O.topk_accuracy(O.clip_probabilities(est, cand), torch.arange(70), torch.arange(50), 10)
    assert acc["top10"] == pytest.approx(ref_acc)


@pytest.mark.parametrize("T,clip", [(360, True), (360, False), (361, False)])
def test_scale_reject_front_end(H, T, clip):
    """Fused ScaleReject (bm/norm.py:239-275,311-345) vs the oracle: bit-exact scaling (same fp32
    op order), identical keep mask, identical compaction."""
    from brainmagick_amd import synthetic
    from brainmagick_amd.norm import DeviceBatchScaler, ScaleReject
    g = _gen(T + clip)
    B, C, Fd, R = 12, 30, 7, 3
    sb = synthetic.make_batch(B, C, T, Fd, 4, seed=T, n_layouts=R)
    meg = sb.meg * 3 + 0.5
    meg[2, 5, 17] = 500.0            # one outlier segment
    meg[7, 0, 0] = -300.0
    sb.meg = meg
    center = torch.randn(R, C, generator=g) * 0.3
    scale = torch.rand(R, C, generator=g) + 0.5
    fcenter = torch.randn(Fd, generator=g)
    fscale = torch.rand(Fd, generator=g) + 0.5
    ref_meg, ref_feat, ref_keep = O.scale_reject(sb.meg, sb.features, sb.recording_index, center,
                                                 scale, fcenter, fscale, limit=20, clip=clip)
    sr = ScaleReject(DeviceBatchScaler(center, scale, fcenter, fscale), limit=20, clip=clip)
    out, keep = sr(sb.to("cuda"))
    assert torch.equal(keep.cpu(), ref_keep)
    assert torch.equal(out.meg.cpu(), ref_meg)
    assert torch.equal(out.features.cpu(), ref_feat)
    assert len(out._recordings) == int(ref_keep.sum())
    if clip:
        assert ref_keep.all() and out.meg.abs().max().item() <= 20
    else:
        n_rej = int((~ref_keep).sum())
        assert n_rej >= 2 and not ref_keep[2] and not ref_keep[7]
        assert sr.rejection_rate == pytest.approx(n_rej / B)


@pytest.mark.parametrize("tag,clip", [("clip", True), ("reject", False)])
def test_scale_reject_against_reference_golden(H, tag, clip):
    from helpers import Golden
    from brainmagick_amd import synthetic
    from brainmagick_amd.norm import DeviceBatchScaler, ScaleReject
    g = Golden("scale_reject")
    meg, feats, rec = g.t("in/meg"), g.t("in/features"), g.t("in/recording_index")
    B, C, T = meg.shape
    recs = [synthetic.Recording(int(r), torch.rand(C, 2)) for r in rec]
    sb = synthetic.SegmentBatch(meg, feats, torch.ones(B, 1, T, dtype=torch.bool),
                                torch.zeros(B, dtype=torch.int64), rec, recs).to("cuda")
    sr = ScaleReject(DeviceBatchScaler(g.t("in/meg_center"), g.t("in/meg_scale"),
                                       g.t("in/feature_center"), g.t("in/feature_scale")),
                     limit=20, clip=clip)
    out, keep = sr(sb)
    assert torch.equal(keep.cpu(), g.t(f"{tag}/keep"))
    assert torch.equal(out.meg.cpu(), g.t(f"{tag}/meg"))
    assert torch.equal(out.features.cpu(), g.t(f"{tag}/features"))


@pytest.mark.parametrize("B,Bc", [(6, 6), (5, 11)])
def test_clip_loss_candidate_gradients(B, Bc):
    """ClipLoss backward w.r.t. BOTH estimates and candidates (learnable feature model)."""
    from brainmagick_amd.losses import ClipLoss
    g = _gen(B * Bc)
    est = torch.randn(B, 9, 40, generator=g, dtype=torch.float64, requires_grad=True)
    cand = (torch.randn(Bc, 9, 40, generator=g, dtype=torch.float64) * 1.5).requires_grad_(True)
    (O.clip_loss(est, cand) * 1.7).backward()
    eg = est.detach().float().cuda().requires_grad_(True)
    cg = cand.detach().float().cuda().requires_grad_(True)
    loss = ClipLoss().cuda()(eg, cg, torch.ones(B, 1, 40, dtype=torch.bool, device="cuda"))
    (loss * 1.7).backward()
    assert rel_l2(eg.grad, est.grad) < GRAD_TOL
    assert rel_l2(cg.grad, cand.grad) < GRAD_TOL


@pytest.mark.parametrize("B,Bc,off", [(48, 48, 0), (40, 100, 0), (32, 128, 64), (256, 256, 0), (300, 301, 1)])
def test_symmetric_clip_loss(B, Bc, off):
    """The opt-in column term (``ClipLoss(symmetric=True)``: "row/col softmax" of the hot-path contract; the
    reference has the row term only): loss and gradients w.r.t. estimates AND candidates against the fp64
    restatement, with extra negatives and with the targets at an offset (a rank's block of gathered candidates)."""
    from brainmagick_amd.losses import ClipLoss
    g = _gen(B + 3 * Bc + off)
    est = torch.randn(B, 7, 30, generator=g, dtype=torch.float64, requires_grad=True)
    cand = (torch.randn(Bc, 7, 30, generator=g, dtype=torch.float64) * 1.5).requires_grad_(True)
    with torch.no_grad():
        est += 0.05 * cand[off:off + B]
    ref = O.clip_loss_symmetric(est, cand, target_offset=off)
    (ref * 1.3).backward()
    eg = est.detach().float().cuda().requires_grad_(True)
    cg = cand.detach().float().cuda().requires_grad_(True)
    mask = torch.ones(B, 1, 30, dtype=torch.bool, device="cuda")
    loss = ClipLoss(symmetric=True).cuda()(eg, cg, mask, target_offset=off)
    (loss * 1.3).backward()
    assert abs(float(loss) - float(ref)) < 1e-5, (float(loss), float(ref))
    assert rel_l2(eg.grad, est.grad) < GRAD_TOL
    assert rel_l2(cg.grad, cand.grad) < GRAD_TOL
    # and the default stays the reference's row term
    rows = ClipLoss().cuda()(eg.detach(), cg.detach(), mask, target_offset=off)
    ref_rows = F.cross_entropy(O.clip_scores(est.detach(), cand.detach()), torch.arange(B) + off)
    assert abs(float(rows) - float(ref_rows)) < 1e-5


@pytest.mark.parametrize("world,rank,B", [(1, 0, 12), (3, 1, 8), (2, 1, 136)])
def test_symmetric_clip_loss_over_gathered_estimates(world, rank, B):
    """``ClipLoss(symmetric=True)`` with ``estimate_all`` (whole-node negatives on both sides: the column term of a
    rank's target candidates runs over EVERY rank's estimates): loss and the gradient w.r.t. all estimate blocks against
    the fp64 restatement; at world 1 it equals the local symmetric loss.  (136: the wide score kernel.)"""
    from brainmagick_amd.losses import ClipLoss
    g = _gen(world * 100 + rank * 10 + B)
    Fd, T = 6, 40
    cand = torch.randn(world * B, Fd, T, generator=g, dtype=torch.float64) * 1.5
    blocks = [(torch.randn(B, Fd, T, generator=g, dtype=torch.float64) + 0.05 * cand[r * B:(r + 1) * B]).requires_grad_(True)
              for r in range(world)]
    ref = O.clip_loss_symmetric_node(torch.cat(blocks), cand, rank, B)
    (ref * 0.7).backward()
    dev = [b.detach().float().cuda().requires_grad_(True) for b in blocks]
    est_all = torch.cat(dev)
    mask = torch.ones(B, 1, T, dtype=torch.bool, device="cuda")
    loss = ClipLoss(symmetric=True).cuda()(dev[rank], cand.float().cuda(), mask, target_offset=rank * B,
                                           estimate_all=est_all)
    (loss * 0.7).backward()
    assert abs(float(loss) - float(ref)) < 1e-5, (float(loss), float(ref))
    for r in range(world):
        assert rel_l2(dev[r].grad, blocks[r].grad) < GRAD_TOL, (r, rel_l2(dev[r].grad, blocks[r].grad))
    if world == 1:
        local = ClipLoss(symmetric=True).cuda()(dev[0].detach(), cand.float().cuda(), mask)
        assert abs(float(local) - float(loss)) < 1e-5


@pytest.mark.parametrize("n_neg", [None, 40])
def test_word_level_wer_batched(H, n_neg):
    """retrieval.get_wer (one GEMM + row kernels) vs the reference's per-segment loop (bm/wer.py:91-120)."""
    from brainmagick_amd import retrieval
    from brainmagick_amd.losses import ClipLoss
    g = _gen(31)
    n, Fd, T = 60, 5, 24
    outputs = torch.randn(n, Fd, T, generator=g)
    estimates = 0.12 * outputs + torch.randn(n, Fd, T, generator=g)
    word_hashes = torch.randint(1, 13, (n,), generator=g).int()
    gen = torch.Generator().manual_seed(5)
    kept = torch.randperm(n, generator=gen)[:n_neg] if n_neg else torch.arange(n)
    ref = O.get_wer_loop(estimates, outputs, word_hashes, kept, topx=3)
    gen = torch.Generator().manual_seed(5)
    got = retrieval.get_wer(ClipLoss().cuda(), estimates, outputs, word_hashes, n_negatives=n_neg,
                            topx=3, generator=gen, batch_size=16)
    assert got["wer"] == pytest.approx(ref["wer"], abs=1e-9)
    assert got["wer_vocab"] == pytest.approx(ref["wer_vocab"], abs=1e-9)
    assert 0 < ref["wer"] < 1


@pytest.mark.gpu
def test_gelu_and_its_derivative_pointwise(H):
    """nn.GELU() (exact erf form, bm/models/simpleconv.py:85-86) and its derivative, element by element against
    fp64: with a faithful fp32 erf the only error left is that of the reference's own fp32 formula
    0.5 x (1 + erf(x / sqrt 2)) -- about one ulp of |x|."""
    x = torch.cat([torch.linspace(-9, 9, 36001), torch.tensor([0.0, -0.0, 1e-30, -1e-20, 0.927734375 * 2 ** 0.5,
                                                                  -0.927734375 * 2 ** 0.5, 30.0, -30.0])])
    n = x.numel() - x.numel() % 4
    x = x[:n].view(1, 1, n).contiguous()
    y = H.affine_act_res(x.cuda(), None, None, None, H.ACT_GELU).double().cpu()
    xd = x.double()
    ref = 0.5 * xd * (1 + torch.erf(xd / 2 ** 0.5))
    assert bool(((y - ref).abs() <= 1.3e-7 * xd.abs().clamp(min=1e-30)).all()), ((y - ref).abs() / xd.abs().clamp(min=1e-30)).max()
    ones = torch.ones_like(x).cuda()
    dy, _, _, _ = H.act_bn_bwd(ones, x.cuda(), None, None, None, None, False, H.ACT_GELU, want_dbias=False)
    dref = 0.5 * (1 + torch.erf(xd / 2 ** 0.5)) + xd * torch.exp(-0.5 * xd * xd) / (2 * torch.pi) ** 0.5
    assert float((dy.double().cpu() - dref).abs().max()) < 4e-7


@pytest.mark.gpu
def test_retrieval_rules_against_reference_fixture(H):
    """The batched GPU retrieval evaluation against numbers produced by the reference's own functions
    (scripts/run_eval_probs.py:237-264, bm/wer.py:82-121 executed from source: tests/golden/retrieval_rules.npz)."""
    import json
    from helpers import Golden
    from brainmagick_amd import retrieval
    from brainmagick_amd.losses import ClipLoss
    g = Golden("retrieval_rules")
    probs = g.t("acc/probs").cuda()
    for k in (1, 5, 10):
        got = retrieval.get_accuracy_from_probs(probs, g.t("acc/target_labels"), g.t("acc/vocab_labels"), topk=k)
        assert got == pytest.approx(float(g.raw[f"acc/top{k}"]), abs=1e-9)
    meta = json.loads(str(g.raw["wer/meta"]))
    gen = torch.Generator().manual_seed(meta["perm_seed"])
    got = retrieval.get_wer(ClipLoss().cuda(), g.t("wer/estimates"), g.t("wer/outputs"), g.t("wer/word_hashes"),
                            n_negatives=meta["n_negatives"], topx=meta["topx"], generator=gen, batch_size=16)
    assert got["wer"] == pytest.approx(float(g.raw["wer/wer"]), abs=1e-9)
    assert got["wer_vocab"] == pytest.approx(float(g.raw["wer/wer_vocab"]), abs=1e-9)


def test_scores_contraction_with_an_operand_over_2gb(H):
    """Whole-node negatives with wav2vec-sized features: 2 048 candidates x (F*T) samples exceed the 2 GB a
    32-bit buffer offset can address; the f32x3 entry point must route that call to the 64-bit-addressing
    kernel instead of wrapping around."""
    M, Cn, K = 32, 2048, 270_016
    g = torch.Generator(device="cuda").manual_seed(7)
    est = torch.randn(M, K, device="cuda", generator=g)
    cand = torch.randn(Cn, K, device="cuda", generator=g)
    assert cand.numel() * 4 > 2 ** 31
    out = H.gemm_nt(est, cand, 1, M, Cn, K, a_strides=(0, K), x_strides=(0, K))[0, :, :, 0]
    ref = est.double() @ cand.double().t()
    assert rel_l2(out, ref) < 2e-6
    del est, cand, ref
    torch.cuda.empty_cache()
