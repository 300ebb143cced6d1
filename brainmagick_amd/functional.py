"""autograd glue: each op of the SimpleConv + ClipLoss hot path as a ``torch.autograd.Function``
whose forward AND backward are libbmhip kernels (``Solver`` calls ``loss.backward()``,
bm/solver.py:385, so the HIP backward kernels must hang off the autograd tape).

No function here touches the CPU oracle or a torch compute op for the hot path; torch is the
tape, the allocator and the stream.
"""

import torch

from . import hip_ops as H



ACT_CODES = {"none": H.ACT_NONE, "gelu": H.ACT_GELU, "relu": H.ACT_RELU, "leaky": H.ACT_LEAKY}


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


_arange_cache = {}


def _arange_i32(n: int, device, repeat: int = 1) -> torch.Tensor:
    """torch.arange(n, int32).repeat(repeat) on `device`, built once per (n, repeat, device): the composed front end
    asked for four of them per step (four tiny launches each time); they are constants.

    READ-ONLY: the returned tensor is shared by every caller (kernel index arguments, autograd contexts); an in-place
    write by a consumer would corrupt every later step.  It is complete before it is handed out (one synchronisation
    when it is built), so it is valid on any stream."""
    key = (n, repeat, device.type, device.index)
    t = _arange_cache.get(key)
    if t is None:
        if len(_arange_cache) > 64:
            _arange_cache.clear()
        t = torch.arange(n, dtype=torch.int32, device=device)
        if repeat != 1:
            t = t.repeat(repeat)
        if device.type == "cuda":
            torch.cuda.current_stream(device).synchronize()
        _arange_cache[key] = t
    return t


def _conv_weight_grads(dy, x, weight, KS, dil, transposed_weight):
    """dW in the layout of the parameter: Conv1d [M,Cin,KS] or ConvTranspose1d(k=1) [Cin,M,1]; written straight into
    the optimizer's flat gradient bucket when the parameter has a registered destination (hip_ops.grad_destination)."""
    dst = H.grad_destination(weight)
    B, M, T = dy.shape
    Cin = x.shape[1]
    # (a registered destination is returned as a FRESH view: autograd only adopts a gradient tensor nobody else holds)
    if transposed_weight:
        out = dst if dst is not None else torch.empty(weight.shape, device=dy.device, dtype=torch.float32)   # [Cin, M, 1]
        # dW[c][m] = sum x[c][t] dy[m][t]: with x as the ROW operand the result lands in the parameter's own
        # [Cin, M] layout, and the head's 640 x 120 gradient fits the wide f16x2 tiles (640 rows x 128 columns)
        # instead of falling to the narrow kernels as 120 rows x 640 columns
        H.gemm_nt(x, dy, B, Cin, M, T, 1, 1, out=out.view(1, Cin, M, 1))
        return out.view(weight.shape) if dst is not None else out
    if dst is not None:
        H.gemm_nt(dy, x, B, M, Cin, T, KS, dil, out=dst.view(1, M, Cin, KS))
        return dst.view(weight.shape)
    return H.gemm_nt(dy, x, B, M, Cin, T, KS, dil).view(weight.shape)


class Conv1dFn(torch.autograd.Function):
    """nn.Conv1d ("same" padding, stride 1) or ConvTranspose1d(k=1) + bias [+ activation].

    Replaces F.conv1d at bm/models/simpleconv.py:113-120 (initial_linear), :185-189 (final head)
    and bm/models/common.py:113-114 for layers without BatchNorm."""

    @staticmethod
    def forward(ctx, x, weight, bias, dil: int, act: int, leak: float, transposed_weight: bool):
        x, weight = _c(x), _c(weight)
        if transposed_weight:
            Cin, M, KS = weight.shape
            assert KS == 1, "ConvTranspose1d is only supported with kernel 1 (simpleconv.py:189)"
            wp = H.pack_weights(weight, 1, M, Cin, 1, 0, 1, M, 0, shape=(x.shape[2], dil))
        else:
            M, Cin, KS = weight.shape
            wp = H.pack_conv_fwd(weight, (x.shape[2], dil))
        need_pre = act != H.ACT_NONE and (x.requires_grad or weight.requires_grad)
        if need_pre:
            # training: the pre-activation is saved anyway; the activation as a streaming pass over it is cheaper
            # than the conv kernel's general epilogue (erf per accumulator element with the matrix cores idle)
            pre, _, _ = H.conv_nn(x, wp, M, KS, dil, bias=bias, want_pre=True, want_out=False)
            out = H.affine_act_res(pre, None, None, None, act, leak)
        else:
            pre, out, _ = H.conv_nn(x, wp, M, KS, dil, bias=bias, act=act, leak=leak)
        ctx.save_for_backward(x, weight, pre)
        ctx.cfg = (dil, act, leak, transposed_weight, KS, bias is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, pre = ctx.saved_tensors
        dil, act, leak, transposed_weight, KS, has_bias = ctx.cfg
        dout = _c(dout)
        dbias = None
        if act != H.ACT_NONE:
            dy, _, _, dbias = H.act_bn_bwd(dout, pre, None, None, None, None, False, act, leak,
                                           want_dbias=has_bias)
        else:
            dy = dout
            if has_bias:
                dbias = H.channel_sum(dy)
        dw = _conv_weight_grads(dy, x, weight, KS, dil, transposed_weight) \
            if ctx.needs_input_grad[1] else None
        dx = None
        if ctx.needs_input_grad[0]:
            if transposed_weight:
                Cin, M, _ = weight.shape
                wp = H.pack_weights(weight, 1, Cin, M, 1, 0, M, 1, 0, shape=(dy.shape[2], dil))
            else:
                M, Cin, _ = weight.shape
                wp = H.pack_conv_dgrad(weight, (dy.shape[2], dil))
            _, dx, _ = H.conv_nn(dy, wp, Cin, KS, dil)
        return dx, dw, dbias, None, None, None, None


class ConvBNActFn(torch.autograd.Function):
    """One ConvSequence layer: Conv1d -> BatchNorm1d -> activation [-> + input] as fused HIP
    kernels (bm/models/common.py:113-119 + :146-147).

    train: conv_nn (the wide f16x2 kernel adds the per-tile sums / sums of squares in its epilogue; the other
    kernels are followed by a channel_stats streaming pass) -> bn_finalize (also updates the running statistics
    like torch) -> affine_act_res.  eval: ONE conv_nn launch with the affine,
    activation and residual folded in its epilogue."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, num_batches,
                training: bool, dil: int, act: int, leak: float, residual: bool, momentum: float,
                eps: float):
        x, weight = _c(x), _c(weight)
        M, Cin, KS = weight.shape
        B, _, T = x.shape
        wp = H.pack_conv_fwd(weight, (T, dil))
        res = x if residual else None
        needs_grad = x.requires_grad or weight.requires_grad
        if training:
            if getattr(wp, "_bm_mode", "") == "f16x2":     # the wide f16x2 conv adds the partial sums in its epilogue
                pre, _, stats = H.conv_nn(x, wp, M, KS, dil, bias=bias, want_pre=True,
                                          want_out=False, want_stats=True)
            else:
                pre, _, _ = H.conv_nn(x, wp, M, KS, dil, bias=bias, want_pre=True, want_out=False)
                stats = H.channel_stats(pre)
            mean, invstd, scale, shift = H.bn_finalize(stats, B * T, gamma, beta, running_mean,
                                                       running_var, num_batches, momentum, eps)
            out = H.affine_act_res(pre, scale, shift, res, act, leak)
        else:
            mean, invstd, scale, shift = H.bn_eval_affine(gamma, beta, running_mean, running_var, eps)
            pre, out, _ = H.conv_nn(x, wp, M, KS, dil, bias=bias, scale=scale, shift=shift, res=res,
                                    act=act, leak=leak, want_pre=needs_grad)
        ctx.save_for_backward(x, weight, pre, scale, shift, mean, invstd)
        ctx.cfg = (training, dil, act, leak, residual, KS, bias is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, pre, scale, shift, mean, invstd = ctx.saved_tensors
        training, dil, act, leak, residual, KS, has_bias = ctx.cfg
        dout = _c(dout)
        M, Cin, _ = weight.shape
        dy, dgamma, dbeta, dbias = H.act_bn_bwd(dout, pre, scale, shift, mean, invstd, training, act,
                                                leak, want_affine_grads=True, want_dbias=has_bias)
        dw = _conv_weight_grads(dy, x, weight, KS, dil, False) if ctx.needs_input_grad[1] else None
        dx = None
        if ctx.needs_input_grad[0]:
            # dx feeds the previous layer's elementwise backward kernel: nobody needs its maximum
            _, dx, _ = H.conv_nn(dy, H.pack_conv_dgrad(weight, (dy.shape[2], dil)), Cin, KS, dil,
                                 res=dout if residual else None, publish_amax=False)
        return (dx, dw, dbias, dgamma, dbeta) + (None,) * 10


class GLUConvFn(torch.autograd.Function):
    """Conv1d(C -> 2C, k = 1 + 2*glu_context) followed by GLU(dim=1) (bm/models/common.py:133-138)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x, weight = _c(x), _c(weight)
        M, Cin, KS = weight.shape
        u, _, _ = H.conv_nn(x, H.pack_conv_fwd(weight, (x.shape[2], 1)), M, KS, 1, bias=bias, want_pre=True,
                            want_out=False)
        out = H.glu_fwd(u)
        ctx.save_for_backward(x, weight, u)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, u = ctx.saved_tensors
        M, Cin, KS = weight.shape
        du, dbias = H.glu_bwd(_c(dout), u, want_dbias=ctx.has_bias)
        dw = _conv_weight_grads(du, x, weight, KS, 1, False) if ctx.needs_input_grad[1] else None  
        dx = None
        if ctx.needs_input_grad[0]:
            _, dx, _ = H.conv_nn(du, H.pack_conv_dgrad(weight, (du.shape[2], 1)), Cin, KS, 1, publish_amax=False)
        return dx, dw, dbias


class SubjectLayersFn(torch.autograd.Function):
    """out[b] = W[subject[b]]^T x[b] (bm/models/common.py:55-58) as a grouped MFMA GEMM: no
    [B, C, D] weight gather is materialised, the weight gradient is a deterministic grouped
    reduction over the segments of each subject."""

    @staticmethod
    def forward(ctx, x, weights, subjects):
        x, weights = _c(x), _c(weights)
        S, C, D = weights.shape
        widx = H.index_i32(_c(subjects.to(torch.int64)), S)      # range-checked like the reference's gather
        wp = H.pack_weights(weights, S, D, C, 1, C * D, 1, D, 0, shape=(x.shape[2], 1))
        _, out, _ = H.conv_nn(x, wp, D, 1, 1, widx=widx)
        ctx.save_for_backward(x, weights, subjects, widx)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weights, subjects, widx = ctx.saved_tensors
        S, C, D = weights.shape
        dout = _c(dout)
        B, _, T = x.shape
        dx = None
        if ctx.needs_input_grad[0]:
            wpt = H.pack_weights(weights, S, C, D, 1, C * D, D, 1, 0, shape=(T, 1))
            _, dx, _ = H.conv_nn(dout, wpt, C, 1, 1, widx=widx)
        dw = None
        if ctx.needs_input_grad[1]:
            order, seg = H.group_by_index(_c(subjects.to(torch.int64)), S)
            dw = torch.empty_like(weights)
            H.gemm_nt(dout, x, B, D, C, T, 1, 1, order=order, seg=seg, G=S, out=dw,
                      out_strides=(C * D, 1, D, 0))
        return dx, dw, None


class ChannelMergerFn(torch.autograd.Function):
    """Spatial attention over sensors (bm/models/common.py:334-358), computed once per distinct
    sensor layout in the batch: positions_u [U, C, 2], layout_index [B] -> which layout a segment
    uses.  ``ban`` = (centre tensor [2], radius) for the training-time sensor dropout (:342-346).
    ``heads`` is [O, D] (shared, the paper's configuration) or [U, O, D]: one set of heads per "layout" -- how
    ``merger_per_subject`` runs, with one entry per (layout, subject) pair of the batch."""

    @staticmethod
    def forward(ctx, meg, heads, positions_u, layout_index, ban_center, ban_radius: float):
        meg, heads, positions_u = _c(meg), _c(heads), _c(positions_u)
        U, C, _ = positions_u.shape
        O, D = heads.shape[-2:]
        per_layout_heads = heads.dim() == 3
        assert not per_layout_heads or heads.shape[0] == U
        emb = H.fourier_emb(positions_u, D)
        seg = _arange_i32(U + 1, meg.device)
        scores = H.gemm_nt(heads, emb, U, O, C, D, a_strides=(O * D if per_layout_heads else 0, D),
                           x_strides=(C * D, D), seg=seg, G=U, force_f32=True).view(U, O, C)
        weights = H.masked_softmax(scores, positions_u, ban_center, ban_radius)
        widx = H.index_i32(_c(layout_index.to(torch.int64)), U)
        wp = H.pack_weights(weights, U, O, C, 1, O * C, C, 1, 0, shape=(meg.shape[2], 1))
        _, out, _ = H.conv_nn(meg, wp, O, 1, 1, widx=widx)
        ctx.save_for_backward(meg, emb, weights, layout_index, widx)
        ctx.dims = (U, C, O, D)
        ctx.per_layout_heads = per_layout_heads
        return out

    @staticmethod
    def backward(ctx, dout):
        meg, emb, weights, layout_index, widx = ctx.saved_tensors
        U, C, O, D = ctx.dims
        dout = _c(dout)
        B, _, T = meg.shape
        dheads = None
        if ctx.needs_input_grad[1]:
            order, seg = H.group_by_index(_c(layout_index.to(torch.int64)), U)
            dweights = H.gemm_nt(dout, meg, B, O, C, T, 1, 1, order=order, seg=seg, G=U).view(U, O, C)
            dscores = H.softmax_bwd(weights, dweights)
            wp = H.pack_weights(dscores, U, O, C, 1, O * C, C, 1, 0, shape=(D, 1))
            uidx = _arange_i32(U, meg.device)
            _, per_layout, _ = H.conv_nn(emb, wp, O, 1, 1, widx=uidx)        # [U, O, D]
            if ctx.per_layout_heads:
                dheads = per_layout
            else:
                dheads = H.sum_over_batch(per_layout) if U > 1 else per_layout[0]
        dmeg = None
        if ctx.needs_input_grad[0]:
            wpt = H.pack_weights(weights, U, C, O, 1, O * C, 1, C, 0, shape=(T, 1))
            _, dmeg, _ = H.conv_nn(dout, wpt, C, 1, 1, widx=widx)
        return dmeg, dheads, None, None, None, None


class FusedFrontEndFn(torch.autograd.Function):
    """ChannelMerger apply -> initial 1x1 conv -> SubjectLayers (bm/models/common.py:355-358,
    bm/models/simpleconv.py:113-120, bm/models/common.py:55-58) as ONE grouped 1x1 conv.

    In the paper's configuration nothing non-linear sits between the three maps, so for a segment of layout u and
    subject s      y = Ws[s]^T (W1 (Wm[u] x) + b1) = Wc[u,s] x + bc[u,s],   Wc = Ws[s]^T W1 Wm[u].
    The composed [D, C] matrices (one per (layout, subject) pair, U*S of them: 27 for one MEG system) cost
    ~1 GFLOP per step; in exchange the two [B, 270, T] intermediates (99.5 MB each at B = 256) are never written
    or read, two of the three forward convs, both data-gradient convs and two of the three weight-gradient
    contractions disappear.  Backward: ONE grouped weight-gradient contraction G[u,s] = sum_{b in (u,s)} dy_b x_b^T
    over the batch, then the chain rule through the small matrices:
        dWs[s] = sum_u [P[u] | b1] [G | cs]^T,   A[u,s] = Ws[s] [G | cs],   dW1 = sum_u (sum_s A) Wm[u]^T,
        db1 = sum_{u,s} A[:, C],   dWm[u] = W1^T sum_s A[u,s]   (then the softmax / logits backward of the merger)
    with P[u] = W1 Wm[u] and cs[u,s] = sum_{b in (u,s), t} dy_b (the bias column, carried as column C of the
    augmented matrices).  The small products run on the fp32-accurate narrow kernels in every fp32-class mode.
    Mathematically identical to the three-layer chain; the rounding differs at the fp32 level (re-association)."""

    @staticmethod
    def forward(ctx, meg, heads, w1, b1, ws, positions_u, layout_index, subjects, ban_center, ban_radius: float):
        meg, heads, w1, ws, positions_u = _c(meg), _c(heads), _c(w1), _c(ws), _c(positions_u)
        U, C, _ = positions_u.shape
        O, Dp = heads.shape
        L = w1.shape[0]
        S, _, D = ws.shape
        B, _, T = meg.shape
        dev = meg.device
        # spatial-attention weights per layout (as ChannelMergerFn)
        emb = H.fourier_emb(positions_u, Dp)
        seg_u = _arange_i32(U + 1, dev)
        scores = H.gemm_nt(heads, emb, U, O, C, Dp, a_strides=(0, Dp), x_strides=(C * Dp, Dp), seg=seg_u,
                           G=U, force_f32=True).view(U, O, C)
        wm = H.masked_softmax(scores, positions_u, ban_center, ban_radius)              # [U, O, C]
        # P[u] = W1 Wm[u], with b1 as column C
        _, p, _ = H.conv_nn(wm, H.pack_conv_fwd(w1), L, 1, 1)                           # [U, L, C]
        bcol = (b1 if b1 is not None else torch.zeros(L, device=dev)).view(1, L, 1).expand(U, L, 1)
        xp = torch.cat([p, bcol], dim=2).repeat_interleave(S, dim=0).contiguous()       # [U*S, L, C+1], pair = u*S + s
        widx_p = _arange_i32(S, dev, repeat=U)
        # Wc[u,s] = Ws[s]^T [P[u] | b1]
        _, wc, _ = H.conv_nn(xp, H.pack_weights(ws, S, D, L, 1, L * D, 1, D, 0), D, 1, 1, widx=widx_p)   # [U*S, D, C+1]
        bias_c = wc[:, :, C].contiguous()                                               # [U*S, D]
        pair = H.index_i32(_c(layout_index.to(torch.int64)), U) * S + H.index_i32(_c(subjects.to(torch.int64)), S)
        wpc = H.pack_weights(wc, U * S, D, C, 1, D * (C + 1), C + 1, 1, 0, shape=(T, 1))
        _, out, _ = H.conv_nn(meg, wpc, D, 1, 1, widx=pair, bias=bias_c, bias_gstride=D)
        ctx.save_for_backward(meg, emb, wm, xp, ws, w1, pair, widx_p, wc)
        ctx.dims = (U, C, O, Dp, L, S, D)
        ctx.has_b1 = b1 is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        meg, emb, wm, xp, ws, w1, pair, widx_p, wc = ctx.saved_tensors
        U, C, O, Dp, L, S, D = ctx.dims
        dout = _c(dout)
        B, _, T = meg.shape
        dev = meg.device
        P = U * S
        # G[p] = sum_{b in p} dy_b x_b^T and the per-pair time sums of dy as column C
        order, seg = H.group_by_index(pair.to(torch.int64), P)
        gaug = torch.empty(P, D, C + 1, device=dev, dtype=torch.float32)
        H.gemm_nt(dout, meg, B, D, C, T, 1, 1, order=order, seg=seg, G=P, out=gaug,
                  out_strides=(D * (C + 1), C + 1, 1, 0))
        gaug[:, :, C] = H.segment_sum_cols(H.time_sums_t(dout), order, seg).t()
        seg_p = _arange_i32(P + 1, dev)
        dws = dw1 = db1 = dheads = None
        if ctx.needs_input_grad[4]:
            t1 = H.gemm_nt(xp, gaug, P, L, D, C + 1, seg=seg_p, G=P)                    # [P, L, D, 1]
            dws = (H.sum_over_batch(t1.view(U, S * L * D)) if U > 1 else t1).view(S, L, D)
        # A[p] = Ws[s] [G | cs], summed over the subjects of a layout
        _, a, _ = H.conv_nn(gaug, H.pack_weights(ws, S, L, D, 1, L * D, D, 1, 0), L, 1, 1, widx=widx_p)   # [P, L, C+1]
        a_u = torch.stack([H.sum_over_batch(a[u * S:(u + 1) * S]) for u in range(U)])   # [U, L, C+1]
        if ctx.has_b1 and ctx.needs_input_grad[3]:
            db1 = H.sum_over_batch(a_u[:, :, C].contiguous()) if U > 1 else a_u[0, :, C].contiguous()
        if ctx.needs_input_grad[2]:
            dw1 = H.gemm_nt(a_u, wm, U, L, O, C, 1, 1, a_strides=(L * (C + 1), C + 1),
                            x_strides=(O * C, C)).view(L, O, 1)
        if ctx.needs_input_grad[1]:
            _, dwm, _ = H.conv_nn(a_u, H.pack_conv_dgrad(w1), O, 1, 1)                  # [U, O, C+1]
            dscores = H.softmax_bwd(wm, dwm[:, :, :C].contiguous())
            wp = H.pack_weights(dscores, U, O, C, 1, O * C, C, 1, 0, shape=(Dp, 1))
            uidx = _arange_i32(U, dev)
            _, per_layout, _ = H.conv_nn(emb, wp, O, 1, 1, widx=uidx)                  # [U, O, Dp]
            dheads = H.sum_over_batch(per_layout) if U > 1 else per_layout[0]
        dmeg = None
        if ctx.needs_input_grad[0]:
            wpt = H.pack_weights(wc, P, C, D, 1, D * (C + 1), 1, C + 1, 0, shape=(T, 1))
            _, dmeg, _ = H.conv_nn(dout, wpt, C, 1, 1, widx=pair)
        return dmeg, dheads, dw1, db1, dws, None, None, None, None, None


# The wide kernels address one operand with 32-bit byte offsets and the conv stages a [Cin, T] window per segment:
# a candidate set beyond 1 GB (2 048 wav2vec2-sized candidates gathered from 8 GPUs = 3 GB) is walked in row
# blocks -- which are also the blocks the per-rank gather delivers.
_CLIP_BLOCK_BYTES = 0x3f000000


def _candidate_blocks(Bc: int, K: int):
    """[(first row, rows)] such that every block's [rows, K] fp32 slab stays below _CLIP_BLOCK_BYTES."""
    if Bc * K * 4 < _CLIP_BLOCK_BYTES:
        return [(0, Bc)]
    rows = max(1, _CLIP_BLOCK_BYTES // (K * 4))
    if rows >= 128:
        rows -= rows % 128
    return [(r0, min(rows, Bc - r0)) for r0 in range(0, Bc, rows)]


def _clip_raw_scores(estimate, candidate, B, Bc, K):
    """Split-K partial tiles [nsplit, B, Bc] of est . cand^T (one launch), or -- for a candidate set walked in
    blocks -- the folded products as a [1, B, Bc] "partial" (one launch + fold per block)."""
    blocks = _candidate_blocks(Bc, K)
    if len(blocks) == 1:
        return H.gemm_nt_partials(estimate, candidate, 1, B, Bc, K, (0, K), (0, K))
    raw = torch.empty(1, B, Bc, device=estimate.device, dtype=torch.float32)
    cand2 = candidate.view(Bc, K)
    for r0, rows in blocks:
        blk = H.share_amax(candidate, cand2[r0:r0 + rows])
        H.gemm_nt(estimate, blk, 1, B, rows, K, 1, 1, a_strides=(0, K), x_strides=(0, K),
                  out=raw.view(-1)[r0:], out_strides=(0, Bc, 1, 0), force_f32=True)
    return raw


class ClipLossFn(torch.autograd.Function):
    """ClipLoss.forward (bm/losses.py:104-114): scores = est . cand^T * inv_norm (split-K MFMA GEMM
    over K = F*T), row-wise cross entropy with the target on the diagonal; the backward is a second
    MFMA GEMM dEst = dScores . cand.  ``symmetric`` (extension, off = the reference): the loss is the mean of the
    row term and the column term (every target candidate classifies the estimates, `bm_clip_ce_cols`)."""

    @staticmethod
    def forward(ctx, estimate, candidate, target_offset: int = 0, col_valid=None, symmetric: bool = False,
                normalize: bool = True):
        """``normalize=False`` (the node-wide column term, losses.ClipLoss): the scores are the plain products
        est . cand -- no candidate norms in the forward pass, no norm correction in the candidates' gradient."""
        estimate, candidate = _c(estimate), _c(candidate)
        B, Bc = estimate.shape[0], candidate.shape[0]
        K = estimate.numel() // B
        assert candidate.numel() // Bc == K
        inv = H.clip_inv_norms(candidate) if normalize else torch.ones(Bc, device=candidate.device, dtype=torch.float32)
        ctx.normalize = normalize
        part = _clip_raw_scores(estimate, candidate, B, Bc, K)
        scores, _, dscaled, loss = H.clip_ce(part, inv, want_grad=True, want_loss=True,
                                             target_offset=target_offset, col_valid=col_valid)
        if symmetric:
            H.clip_ce_cols(scores, inv, dscaled, loss, target_offset=target_offset)
        if ctx.needs_input_grad[1]:
            ctx.save_for_backward(candidate, dscaled, estimate, scores, inv)
        else:
            ctx.save_for_backward(candidate, dscaled)
        ctx.shape = estimate.shape
        ctx.cand_shape = candidate.shape
        ctx.mark_non_differentiable(scores)
        return loss, scores

    @staticmethod
    def backward(ctx, dloss, _dscores):
        candidate, dscaled = ctx.saved_tensors[:2]
        B, Bc = dscaled.shape
        K = candidate.numel() // Bc
        alpha = _c(dloss).view(1)
        blocks = _candidate_blocks(Bc, K)
        cand2 = candidate.view(Bc, K)
        dest = None
        if ctx.needs_input_grad[0]:
            # dEst = sum over candidate blocks of dScores[:, block] . cand[block]; the running sum is the
            # conv epilogue's residual operand, written in place
            for r0, rows in blocks:
                wp = H.pack_weights(dscaled.view(-1)[r0:], 1, B, rows, 1, 0, Bc, 1, 0, alpha=alpha, shape=(K, 1))
                # (views are new tensor objects: hand the candidates' published maximum on instead of re-scanning them)
                blk = H.share_amax(candidate, cand2[r0:r0 + rows].view(1, rows, K))
                _, dest, _ = H.conv_nn(blk, wp, B, 1, 1, res=dest, out=dest)
            dest = H.share_amax(dest, dest.view(ctx.shape))
        dcand = None
        if ctx.needs_input_grad[1]:
            # learnable candidates (DeepMel feature model): dcand_o = sum_b dscaled[b,o] est_b - coef_o cand_o
            estimate, scores, inv = ctx.saved_tensors[2:]
            dcand = torch.empty(Bc, K, device=candidate.device, dtype=torch.float32)
            for r0, rows in blocks:
                wpt = H.pack_weights(dscaled.view(-1)[r0:], 1, rows, B, 1, 0, 1, Bc, 0, alpha=alpha, shape=(K, 1))
                H.conv_nn(estimate.view(1, B, K), wpt, rows, 1, 1,
                          out=dcand[r0:r0 + rows] if len(blocks) > 1 else dcand)
            if ctx.normalize:
                coef = H.clip_cand_coef(dscaled, scores, inv, alpha)
                H.row_axpy_sub(dcand, cand2, coef)
            dcand = dcand.view(ctx.cand_shape)
        return dest, dcand, None, None, None, None


def clip_scores(estimate, candidate, want_probs=False):
    """ClipLoss.get_scores / get_probabilities (bm/losses.py:77-102), no autograd."""
    estimate, candidate = _c(estimate), _c(candidate)
    B, Bc = estimate.shape[0], candidate.shape[0]
    K = estimate.numel() // B
    inv = H.clip_inv_norms(candidate)
    part = _clip_raw_scores(estimate, candidate, B, Bc, K)
    scores, probs, _, _ = H.clip_ce(part, inv, want_probs=want_probs)
    return probs if want_probs else scores


def clip_forward_timed(estimate, candidate, inv, timer):
    """bench.py: the ClipLoss forward (scores contraction + split fold + row softmax / CE + dScores) on
    precomputed candidate norms, optionally with per-kernel HIP events."""
    B, Bc = estimate.shape[0], candidate.shape[0]
    K = estimate.numel() // B
    H.set_kernel_timer(timer)
    try:
        part = _clip_raw_scores(estimate, candidate, B, Bc, K)
        return H.clip_ce(part, inv, want_grad=True, want_loss=True)
    finally:
        H.set_kernel_timer(None)
