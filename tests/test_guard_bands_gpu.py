"""Guard bands: every output / workspace the hand-scheduled kernels write lives inside a canary-bordered,
NaN-poisoned allocation ("sanitizers" row).

``conv_nn_h2w.hip`` and ``gemm_nt_h2w.hip`` issue raw buffer loads, LDS-DMA copies and hand-counted waits; their
stores are bounds-checked by hand.  Here every ``torch.empty`` / ``empty_like`` / ``zeros`` the wrappers of
``brainmagick_amd.hip_ops`` perform while an op runs is served from a larger buffer:

    [ GUARD canary floats | payload (NaN-poisoned unless zero-initialised) | GUARD canary floats ]

After the op: (i) every canary is intact -- no store left its tile; (ii) the results hold no NaN -- no element of
an output was skipped and no stale (never written) workspace element was consumed; (iii) the numbers are right
(fp64 reference).  Shapes are the ragged ones: B = 173, T in {130, 343, 361, 777}, M in {120, 270, 1024}.
"""
import contextlib
import math

import pytest
import torch
from torch.nn import functional as F

from helpers import rel_l2

pytestmark = pytest.mark.gpu

_REAL_EMPTY, _REAL_EMPTY_LIKE, _REAL_ZEROS = torch.empty, torch.empty_like, torch.zeros
GUARD = 1024                    # floats on either side (4 KB: a whole stray 16-byte-per-lane wavefront store)
CANARY = -7.0e37
FWD_TOL, GRAD_TOL = 5e-6, 2e-5


class Arena:
    """Serves the allocations of the code under test from guarded buffers and checks them afterwards."""

    def __init__(self):
        self.blocks = []

    def _alloc(self, shape, dtype, device, poison):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
        n = 1
        for s in shape:
            n *= int(s)
        nbytes = n * _REAL_EMPTY((), dtype=dtype).element_size()
        padded = nbytes + (-nbytes) % 4                 # the upper canaries start on a 4-byte boundary
        guard_bytes = GUARD * 4
        raw = _REAL_EMPTY(padded + 2 * guard_bytes, dtype=torch.uint8, device=device)
        lo = raw[:guard_bytes].view(torch.float32)
        hi = raw[guard_bytes + padded:].view(torch.float32)
        lo.fill_(CANARY)
        hi.fill_(CANARY)
        payload = raw[guard_bytes:guard_bytes + nbytes]
        if poison and dtype == torch.float32:
            payload.view(torch.float32).fill_(float("nan"))
        elif poison:
            payload.fill_(0xFF)                      # NaN for f16 / bf16 pairs, -1 for integers
        else:
            payload.zero_()
        self.blocks.append((raw, lo, hi, nbytes))
        return payload.view(dtype).view(shape)

    @contextlib.contextmanager
    def active(self):
        real_empty, real_empty_like, real_zeros = _REAL_EMPTY, _REAL_EMPTY_LIKE, _REAL_ZEROS
        arena = self

        def empty(*shape, dtype=None, device=None, **kw):
            if device is None or torch.device(device).type != "cuda" or kw.get("pin_memory"):
                return real_empty(*shape, dtype=dtype, device=device, **kw)
            return arena._alloc(shape, dtype or torch.float32, device, poison=True)

        def empty_like(t, **kw):
            if not t.is_cuda or kw:
                return real_empty_like(t, **kw)
            return arena._alloc(tuple(t.shape), t.dtype, t.device, poison=True)

        def zeros(*shape, dtype=None, device=None, **kw):
            if device is None or torch.device(device).type != "cuda":
                return real_zeros(*shape, dtype=dtype, device=device, **kw)
            return arena._alloc(shape, dtype or torch.float32, device, poison=False)

        torch.empty, torch.empty_like, torch.zeros = empty, empty_like, zeros
        try:
            yield self
        finally:
            torch.empty, torch.empty_like, torch.zeros = real_empty, real_empty_like, real_zeros

    def check(self, what):
        torch.cuda.synchronize()
        assert self.blocks, f"{what}: the arena served no allocation (the patch missed the wrappers)"
        for i, (raw, lo, hi, nbytes) in enumerate(self.blocks):
            assert bool((lo == CANARY).all()), f"{what}: allocation {i} ({nbytes} B): a store landed BELOW the buffer"
            assert bool((hi == CANARY).all()), f"{what}: allocation {i} ({nbytes} B): a store landed ABOVE the buffer"


def _no_nan(t, what):
    assert t is not None
    assert not bool(torch.isnan(t).any()), f"{what}: NaN in the result (a skipped element or a stale workspace read)"


@pytest.fixture(scope="module")
def H():
    from brainmagick_amd import hip_ops
    hip_ops.set_compute_dtype("f16x2")
    yield hip_ops
    hip_ops.set_compute_dtype(hip_ops.DEFAULT_COMPUTE_DTYPE)


def _gen(seed):
    return torch.Generator().manual_seed(seed)


# bm_conv1d_nn_h2 ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,Cin,M,KS,dil,T", [
    (173, 64, 320, 3, 2, 130),        # ragged batch, one partly filled 192-column tile
    (5, 320, 270, 3, 16, 343),        # M not a multiple of the 64-row blocks, odd T (dword path), largest halo
    (3, 320, 1024, 1, 1, 361),        # several row tiles, T = 2 tiles - 23
    (2, 48, 120, 3, 4, 777),          # 5 column tiles, the last one holds 9 columns; M below one 128-row tile
    (173, 320, 320, 3, 1, 361),       # the production shape at the ragged batch
])
def test_conv_outputs_stay_inside_their_buffers(H, B, Cin, M, KS, dil, T):
    g = _gen(B + Cin + M + T)
    x = torch.randn(B, Cin, T, generator=g).cuda()
    w = (torch.randn(M, Cin, KS, generator=g) / math.sqrt(Cin * KS)).cuda()
    b = torch.randn(M, generator=g).cuda()
    res = torch.randn(B, M, T, generator=g).cuda()
    arena = Arena()
    with arena.active():
        wp = H.pack_conv_fwd(w, (T, dil))
        assert getattr(wp, "_bm_mode", "") == "f16x2", "shape not covered by the wide f16x2 conv: pick another"
        # the two forms the training step uses: pre-activation + BatchNorm partial sums (forward), and output + residual
        # with the published maximum (data gradient)
        pre, _, stats = H.conv_nn(x, wp, M, KS, dil, bias=b, want_pre=True, want_out=False, want_stats=True)
        _, out, _ = H.conv_nn(x, wp, M, KS, dil, bias=b, res=res)
        amax_out = H.amax(out)            # the maximum the epilogue published (finalize launch included)
    arena.check(f"conv {Cin}->{M} k{KS} d{dil} T={T} B={B}")
    for t, name in ((pre, "pre"), (out, "out"), (stats, "stats"), (amax_out, "amax")):
        _no_nan(t, name)
    ref = F.conv1d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=KS // 2 * dil, dilation=dil)
    assert rel_l2(pre, ref) < FWD_TOL
    assert rel_l2(out, ref + res.double().cpu()) < FWD_TOL
    assert abs(float(amax_out.max()) - float(out.abs().max())) <= 1e-6 * float(out.abs().max())
    # the BatchNorm partial sums of the epilogue: sum and sum of squares per channel over (B, T)
    st = stats.double().cpu()
    st = st.sum(1) if getattr(stats, "_bm_channel_major", False) else st.sum(0)
    assert rel_l2(st[:, 0], ref.sum((0, 2))) < 1e-4
    assert rel_l2(st[:, 1], (ref * ref).sum((0, 2))) < 1e-5


# bm_gemm_nt_h2 / _rows (weight gradients) ----------------------------------------------------------------------------
@pytest.mark.parametrize("B,M,Cn,KS,dil,T", [
    (173, 320, 320, 3, 2, 360),       # ragged batch, flat (segment, time) axis + per-row scales
    (7, 320, 320, 3, 16, 343),        # odd T: per-segment padded stages, no row scales
    (5, 270, 270, 1, 1, 361),         # M within 25 % of 320, odd T
    (3, 1024, 320, 1, 1, 130),        # several row tiles, T barely above one tile
    (2, 320, 64, 3, 1, 777),
])
def test_weight_gradient_outputs_stay_inside_their_buffers(H, B, M, Cn, KS, dil, T):
    g = _gen(B * 3 + M + Cn + T)
    dy = torch.randn(B, M, T, generator=g).cuda()
    x = torch.randn(B, Cn, T, generator=g).cuda()
    if not H.lib().bm_gemm_nt_h2_covers(M, Cn, KS, B, T, 1, dil, 0):
        pytest.skip("shape not covered by the wide f16x2 weight-gradient kernel")
    arena = Arena()
    with arena.active():
        dw = H.gemm_nt(dy, x, B, M, Cn, T, KS, dil)
    arena.check(f"wgrad {M}x{Cn} k{KS} d{dil} T={T} B={B}")
    _no_nan(dw, "dW")
    xs = x.double().cpu()
    ref = torch.zeros(M, Cn, KS, dtype=torch.float64)
    for j in range(KS):
        sh = (j - KS // 2) * dil
        xsh = torch.zeros_like(xs)
        if sh >= 0:
            xsh[:, :, :T - sh] = xs[:, :, sh:]
        else:
            xsh[:, :, -sh:] = xs[:, :, :T + sh]
        ref[:, :, j] = torch.einsum("bmt,bct->mc", dy.double().cpu(), xsh)
    assert rel_l2(dw.view(M, Cn, KS), ref) < GRAD_TOL


def test_grouped_weight_gradient_stays_inside_its_buffers(H):
    """bm_gemm_nt_h2_grouped: groups of unequal size, an empty group, T = 343."""
    g = _gen(17)
    B, M, Cn, T, G = 173, 270, 208, 343, 6
    dy = torch.randn(B, M, T, generator=g).cuda()
    x = torch.randn(B, Cn, T, generator=g).cuda()
    idx = torch.randint(0, G - 1, (B,), generator=g)           # group G-1 stays empty
    arena = Arena()
    with arena.active():
        order, seg = H.group_by_index(idx.cuda(), G)
        dw = H.gemm_nt(dy, x, B, M, Cn, T, 1, 1, order=order, seg=seg, G=G)
    arena.check("grouped wgrad")
    _no_nan(dw, "grouped dW")
    ref = torch.zeros(G, M, Cn, dtype=torch.float64)
    for k in range(G):
        sel = idx == k
        if sel.any():
            ref[k] = torch.einsum("bmt,bct->mc", dy.double().cpu()[sel], x.double().cpu()[sel])
    assert rel_l2(dw.view(G, M, Cn), ref) < GRAD_TOL
    assert float(dw.view(G, M, Cn)[G - 1].abs().max()) == 0.0


# bm_clip_scores_h2 --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,Bc,F_,T", [(173, 173, 120, 343), (256, 301, 120, 130), (130, 1024, 24, 777)])
def test_clip_scores_stay_inside_their_buffers(H, B, Bc, F_, T):
    g = _gen(B + Bc + T)
    K = F_ * T
    est = torch.randn(B, K, generator=g).cuda()
    cand = torch.randn(Bc, K, generator=g).cuda()
    arena = Arena()
    with arena.active():
        inv = H.clip_inv_norms(cand)
        part = H.gemm_nt_partials(est, cand, 1, B, Bc, K, (0, K), (0, K))
        scores, probs, dscaled, loss = H.clip_ce(part, inv, want_probs=True, want_grad=True, want_loss=True)
    arena.check(f"clip scores {B}x{Bc} K={K}")
    for t, name in ((part, "partial tiles"), (scores, "scores"), (probs, "probs"), (dscaled, "dscaled"), (loss, "loss")):
        _no_nan(t, name)
    c = cand.double().cpu()
    ref = est.double().cpu() @ (c / (1e-8 + c.norm(dim=1, keepdim=True))).t()
    assert rel_l2(scores, ref) < FWD_TOL
    ref_loss = F.cross_entropy(ref, torch.arange(B))
    assert abs(float(loss) - float(ref_loss)) < 1e-5 * max(1.0, abs(float(ref_loss)))


# bm_act_bn_bwd, bm_glu_bwd, bm_affine_act_res, bm_glu_fwd ---------------------------------------------------------------
@pytest.mark.parametrize("B,C,T", [(173, 120, 130), (7, 270, 343), (5, 320, 361), (3, 1024, 777), (173, 320, 360)])
def test_streaming_kernels_stay_inside_their_buffers(H, B, C, T):
    g = _gen(B + C + T)
    y = torch.randn(B, C, T, generator=g)
    dout = torch.randn(B, C, T, generator=g)
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g)
    res = torch.randn(B, C, T, generator=g)
    yd = y.double()
    mean = yd.mean((0, 2))
    var = yd.var((0, 2), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = gamma.double() * invstd
    shift = beta.double() - mean * scale
    f32 = lambda t: t.float().cuda()    # noqa: E731
    arena = Arena()
    with arena.active():
        out = H.affine_act_res(f32(y), f32(scale), f32(shift), f32(res), H.ACT_GELU)
        dy, dgamma, dbeta, dbias = H.act_bn_bwd(f32(dout), f32(y), f32(scale), f32(shift), f32(mean), f32(invstd),
                                                True, H.ACT_GELU, want_affine_grads=True)
        dy_amax = H.amax(dy)
        rows = H.row_amax_of(dy)
    arena.check(f"bn/act streaming kernels B={B} C={C} T={T}")
    for t, name in ((out, "affine_act_res"), (dy, "dy"), (dgamma, "dgamma"), (dbeta, "dbeta"), (dy_amax, "amax(dy)")):
        _no_nan(t, name)
    # fp64 autograd reference of BatchNorm(train) -> GELU -> (+res)
    yr = yd.clone().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    z = F.batch_norm(yr, None, None, gr, br, True, 0.1, 1e-5)
    o = F.gelu(z) + res.double()
    o.backward(dout.double())
    assert rel_l2(out, o) < FWD_TOL
    assert rel_l2(dy, yr.grad) < GRAD_TOL
    assert rel_l2(dgamma, gr.grad) < GRAD_TOL and rel_l2(dbeta, br.grad) < GRAD_TOL
    assert abs(float(dy_amax.max()) - float(dy.abs().max())) <= 1e-6 * float(dy.abs().max())
    if rows is not None:
        _no_nan(rows, "per-channel maxima")
        assert torch.allclose(rows.cpu(), dy.abs().amax((0, 2)).cpu(), rtol=1e-6, atol=0)
    if rows is not None and C == 320 and T % 4 == 0:
        # the row-scaled weight gradient (per-channel maxima from the producer above), flat (segment, time) axis
        x = torch.randn(B, 320, T, generator=g).cuda()
        arena3 = Arena()
        with arena3.active():
            dw = H.gemm_nt(dy, x, B, C, 320, T, 3, 2)
        arena3.check("row-scaled weight gradient")
        _no_nan(dw, "dW (row scales)")
        xs, dyd = x.double().cpu(), dy.double().cpu()
        for j, sh in enumerate((-2, 0, 2)):
            xsh = torch.zeros_like(xs)
            if sh >= 0:
                xsh[:, :, :T - sh] = xs[:, :, sh:]
            else:
                xsh[:, :, -sh:] = xs[:, :, :T + sh]
            assert rel_l2(dw.view(C, 320, 3)[:, :, j], torch.einsum("bmt,bct->mc", dyd, xsh)) < GRAD_TOL
    if C % 2 == 0:
        u = torch.randn(B, C, T, generator=g)
        dh = torch.randn(B, C // 2, T, generator=g)
        arena2 = Arena()
        with arena2.active():
            h = H.glu_fwd(u.cuda())
            du, dbias_u = H.glu_bwd(dh.cuda(), u.cuda())
            du_amax = H.amax(du)
        arena2.check(f"glu kernels B={B} C={C} T={T}")
        for t, name in ((h, "glu"), (du, "du"), (dbias_u, "dbias"), (du_amax, "amax(du)")):
            _no_nan(t, name)
        ur = u.double().requires_grad_(True)
        hr = F.glu(ur, dim=1)
        hr.backward(dh.double())
        assert rel_l2(h, hr) < FWD_TOL and rel_l2(du, ur.grad) < GRAD_TOL
        assert rel_l2(dbias_u, ur.grad.sum((0, 2))) < GRAD_TOL


def test_a_stray_store_is_caught():
    """The arena itself: a write one element past / before a served buffer trips the check."""
    arena = Arena()
    with arena.active():
        t = torch.empty(33, 7, device="cuda", dtype=torch.float32)
        assert bool(torch.isnan(t).all())
    arena.check("untouched")
    raw = arena.blocks[0][0]
    raw[GUARD * 4 + 33 * 7 * 4:].view(torch.float32)[0] = 1.0      # first float above the payload
    with pytest.raises(AssertionError, match="ABOVE"):
        arena.check("overrun")
