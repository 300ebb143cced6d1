"""Host-side mirror of ``bm/models/common.py`` for the hot path: same class names, constructor
arguments, parameter / buffer names (``state_dict`` keys) and error behaviour as the reference,
with every tensor op replaced by libbmhip kernels (``brainmagick_amd.functional``).

Modules only hold parameters and sequence the HIP ops; they raise on CPU tensors (no fallback).

What is the reference's own text here, said plainly: the three small helpers ``ScaledEmbedding``
(bm/models/common.py:28-42), ``LayerScale`` (:65-76) and ``pad_multiple`` (:22-25) are the reference's code as is
(a few lines each, off the hot path, torch ops only), and the CONSTRUCTORS of ``ConvSequence``, ``ChannelMerger``,
``SubjectLayers`` and ``PositionGetter.get_positions / is_invalid`` follow the reference statement by statement, because
module order, attribute names and the order of the random draws ARE the interoperability contract (``state_dict`` keys,
same-seed initialisation: tests/golden/make_golden.py asserts bit-identical parameters from the same seed).  Every
``forward`` is written over the HIP ops.
"""
from functools import partial
import math
import typing as tp

import torch
from torch import nn

from .. import functional as BF
from .. import hip_ops as H


class ScaledEmbedding(nn.Module):
    """bm/models/common.py:29-43.  Tiny [n_subjects, dim] lookup (host-side plumbing)."""
    def __init__(self, num_embeddings: int, embedding_dim: int, scale: float = 10.):
        super().__init__()
        self.embedding = nn.Embedding(num_embeddings, embedding_dim)
        self.embedding.weight.data /= scale
        self.scale = scale

    @property
    def weight(self):
        return self.embedding.weight * self.scale

    def forward(self, x):
        return self.embedding(x) * self.scale


class SubjectLayers(nn.Module):
    """Per subject linear layer (bm/models/common.py:45-62); parameter ``weights`` [S, C, D]."""
    def __init__(self, in_channels: int, out_channels: int, n_subjects: int, init_id: bool = False):
        super().__init__()
        self.weights = nn.Parameter(torch.randn(n_subjects, in_channels, out_channels))
        if init_id:
            assert in_channels == out_channels
            self.weights.data[:] = torch.eye(in_channels)[None]
        self.weights.data *= 1 / in_channels**0.5

    def forward(self, x, subjects):
        return BF.SubjectLayersFn.apply(x, self.weights, subjects)

    def __repr__(self):
        S, C, D = self.weights.shape
        return f"SubjectLayers({C}, {D}, {S})"


class _Activation(nn.Module):
    """Marker module standing where the reference puts nn.GELU / nn.ReLU / nn.LeakyReLU
    (it keeps the ``nn.Sequential`` indices, hence the state_dict keys, identical); the
    activation itself runs in the epilogue of the preceding HIP op."""
    def __init__(self, kind: str, leak: float = 0.):
        super().__init__()
        self.kind = kind
        self.leak = leak

    @property
    def code(self) -> int:
        return BF.ACT_CODES[self.kind]

    def extra_repr(self):
        return self.kind + (f", {self.leak}" if self.kind == "leaky" else "")


def make_activation(gelu: bool, relu_leakiness: float) -> tp.Callable[[], _Activation]:
    """Activation choice of bm/models/simpleconv.py:85-90."""
    if gelu:
        return partial(_Activation, "gelu")
    if relu_leakiness:
        return partial(_Activation, "leaky", relu_leakiness)
    return partial(_Activation, "relu")


class _GLU(nn.Module):
    """Marker for nn.GLU(dim=1) (fused in GLUConvFn)."""


class LayerScale(nn.Module):
    """bm/models/common.py:65-76 (Touvron et al. 2021): a learnt per-channel factor on the residual branch,
    ``boost * scale``, initialised to ``init``.  Off the hot path (no grid of the paper sets it): a broadcast
    multiplication on the GPU, autograd included."""
    def __init__(self, channels: int, init: float = 0.1, boost: float = 5.):
        super().__init__()
        self.scale = nn.Parameter(torch.zeros(channels, requires_grad=True))
        self.scale.data[:] = init / boost
        self.boost = boost

    def forward(self, x):
        return (self.boost * self.scale[:, None]) * x


class ConvSequence(nn.Module):
    """bm/models/common.py:79-151.  Same constructor; builds the same ``sequence`` / ``glus``
    module lists (so ``sequence.k.0`` is the conv, ``sequence.k.1`` the BatchNorm1d, ``glus.k.0``
    the GLU conv -- and, with the options no grid of the paper uses, the Dropout / rewrite conv / LayerScale /
    post-skip modules at the reference's indices) but runs each layer as fused HIP kernels.

    The hot path is conv [-> BatchNorm] -> activation [-> + input] as ONE fused function.  ``dropout`` /
    ``dropout_input`` (nn.Dropout), ``rewrite`` (1x1 conv + LeakyReLU), ``scale`` (LayerScale) and ``post_skip``
    (a depthwise 1x1 conv without bias = one factor per channel) sit between the activation and the skip
    addition: a layer that has any of them runs the fused function without its residual, the extras as (GPU) torch
    ops or the 1x1 HIP conv, and the addition as one streaming launch -- slower than the hot path, never an error."""

    def __init__(self, channels: tp.Sequence[int], kernel: int = 4, dilation_growth: int = 1,
                 dilation_period: tp.Optional[int] = None, stride: int = 2,
                 dropout: float = 0.0, leakiness: float = 0.0, groups: int = 1,
                 decode: bool = False, batch_norm: bool = False, dropout_input: float = 0,
                 skip: bool = False, scale: tp.Optional[float] = None, rewrite: bool = False,
                 activation_on_last: bool = True, post_skip: bool = False, glu: int = 0,
                 glu_context: int = 0, glu_glu: bool = True, activation: tp.Any = None) -> None:
        super().__init__()
        unsupported = dict(stride=stride != 1, decode=decode)
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(
                f"ConvSequence options {bad} are outside the MI355X hot path (unused by the paper's "
                "grids, SURVEY.md §2.2)")
        if kernel % 2 != 1:
            raise NotImplementedError("only odd kernels ('same' padding) are on the hot path")
        dilation = 1
        channels = tuple(channels)
        self.skip = skip
        self.sequence = nn.ModuleList()
        self.glus = nn.ModuleList()
        if activation is None:
            activation = partial(_Activation, "leaky", leakiness)
        self._plan: tp.List[dict] = []
        for k, (chin, chout) in enumerate(zip(channels[:-1], channels[1:])):
            layers: tp.List[nn.Module] = []
            is_last = k == len(channels) - 2
            plan = dict(dilation=1, act=None, bn=None, glu=None, pre=None, post=[])
            if k == 0 and dropout_input:
                assert 0 < dropout_input < 1
                layers.append(nn.Dropout(dropout_input))
                plan["pre"] = layers[-1]
            if dilation_period and (k % dilation_period) == 0:
                dilation = 1
            pad = kernel // 2 * dilation
            layers.append(nn.Conv1d(chin, chout, kernel, 1, pad, dilation=dilation, groups=groups if k > 0 else 1))
            plan["conv"] = layers[-1]
            plan["dilation"] = dilation
            dilation *= dilation_growth
            if activation_on_last or not is_last:
                if batch_norm:
                    layers.append(nn.BatchNorm1d(num_features=chout))
                    plan["bn"] = layers[-1]
                act = activation()
                layers.append(act)
                plan["act"] = act
                if dropout:
                    layers.append(nn.Dropout(dropout))
                    plan["post"].append(("dropout", layers[-1]))
                if rewrite:
                    layers += [nn.Conv1d(chout, chout, 1), _Activation("leaky", leakiness)]
                    plan["post"].append(("rewrite", layers[-2], layers[-1]))
            if chin == chout and skip:
                if scale is not None:
                    layers.append(LayerScale(chout, scale))
                    plan["post"].append(("scale", layers[-1]))
                if post_skip:
                    layers.append(nn.Conv1d(chout, chout, 1, groups=chout, bias=False))
                    plan["post"].append(("post_skip", layers[-1]))
            self.sequence.append(nn.Sequential(*layers))
            if glu and (k + 1) % glu == 0:
                ch = 2 * chout if glu_glu else chout
                act = _GLU() if glu_glu else activation()
                self.glus.append(
                    nn.Sequential(
                        nn.Conv1d(chout, ch, 1 + 2 * glu_context, padding=glu_context), act))
                plan["glu"] = act
            else:
                self.glus.append(None)
            self._plan.append(plan)

    def _grouped_layer(self, x, plan, code, leak, fused_residual):
        """``groups`` > 1 (off the hot path: no grid of the paper sets it): conv, BatchNorm and activation are all
        per-channel or block-diagonal, so the layer is `groups` independent layers on channel slices -- the same fused
        functions on views of the parameters and buffers (``num_batches_tracked`` counts once)."""
        conv, bn = plan["conv"], plan["bn"]
        G = conv.groups
        cin, cout = conv.in_channels // G, conv.out_channels // G
        outs = []
        # snapshot of the counter BEFORE group 0 increments it: with momentum=None (cumulative average) every group
        # must see the same count
        count0 = bn.num_batches_tracked.clone() if bn is not None and G > 1 else None
        for g in range(G):
            xg = x[:, g * cin:(g + 1) * cin].contiguous()
            w = conv.weight[g * cout:(g + 1) * cout]
            b = conv.bias[g * cout:(g + 1) * cout] if conv.bias is not None else None
            if bn is not None:
                sl = slice(g * cout, (g + 1) * cout)
                count = bn.num_batches_tracked if g == G - 1 else count0.clone()   # the module's counter moves once, last
                outs.append(BF.ConvBNActFn.apply(
                    xg, w, b, bn.weight[sl], bn.bias[sl], bn.running_mean[sl], bn.running_var[sl], count,
                    self.training, plan["dilation"], code, leak, fused_residual, bn.momentum, bn.eps))
            else:
                outs.append(BF.Conv1dFn.apply(xg, w, b, plan["dilation"], code, leak, False))
        return torch.cat(outs, dim=1)

    def forward(self, x: tp.Any) -> tp.Any:
        for module_idx, module in enumerate(self.sequence):
            plan = self._plan[module_idx]
            conv = plan["conv"]
            act = plan["act"]
            code = act.code if act is not None else H.ACT_NONE
            leak = act.leak if act is not None else 0.
            residual = self.skip and conv.in_channels == conv.out_channels
            old_x = x
            if plan["pre"] is not None:
                x = plan["pre"](x)                                  # nn.Dropout on the input of the sequence
            fused_residual = residual and not plan["post"] and plan["pre"] is None
            if conv.groups != 1:
                x = self._grouped_layer(x, plan, code, leak, fused_residual)
            elif plan["bn"] is not None:
                bn = plan["bn"]
                x = BF.ConvBNActFn.apply(
                    x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                    bn.num_batches_tracked, self.training, plan["dilation"], code, leak, fused_residual,
                    bn.momentum, bn.eps)
            else:
                x = BF.Conv1dFn.apply(x, conv.weight, conv.bias, plan["dilation"], code, leak, False)
            for op in plan["post"]:                                 # the off-path extras, in the reference's order
                if op[0] == "dropout":
                    x = op[1](x)
                elif op[0] == "rewrite":
                    x = BF.Conv1dFn.apply(x, op[1].weight, op[1].bias, 1, op[2].code, op[2].leak, False)
                elif op[0] == "scale":
                    x = op[1](x)
                else:                                               # depthwise 1x1 without bias: one factor per channel
                    x = x * op[1].weight.view(1, -1, 1)
            if residual and not (fused_residual and plan["bn"] is not None):
                x = _AddFn.apply(x if x.is_contiguous() else x.contiguous(), old_x)
            glu = self.glus[module_idx]
            if glu is not None:
                gconv = glu[0]
                if isinstance(glu[1], _GLU):
                    x = BF.GLUConvFn.apply(x, gconv.weight, gconv.bias)
                else:
                    x = BF.Conv1dFn.apply(x, gconv.weight, gconv.bias, 1, glu[1].code, glu[1].leak,
                                          False)
        return x


class _AddFn(torch.autograd.Function):
    """x + old_x for the (rare) skip layer without BatchNorm, as an affine_act_res launch."""
    @staticmethod
    def forward(ctx, y, res):
        return H.affine_act_res(y.contiguous(), None, None, res.contiguous(), H.ACT_NONE)

    @staticmethod
    def backward(ctx, dout):
        return dout, dout


class PositionGetter:
    """bm/models/common.py:187-236.  Per-recording 2-D sensor layout, normalised to [0,1]^2;
    sensors missing from the layout (or padded channels) get INVALID = -0.1.

    ``mne`` is imported lazily and only when a recording carries a real ``mne_info``; recordings
    that already expose a ``layout`` tensor (synthetic data, pre-extracted layouts) are used as is.
    MI355X-first addition: ``get_unique_layouts`` de-duplicates the layouts of a batch so the
    attention weights are computed once per layout, not once per segment."""
    INVALID = -0.1

    def __init__(self) -> None:
        self._cache: tp.Dict[int, torch.Tensor] = {}
        self._invalid_names: tp.Set[str] = set()
        self._device_cache: tp.Dict[tp.Any, tp.Tuple[torch.Tensor, torch.Tensor]] = {}
        self._layout_set_cache: tp.Dict[tp.Any, torch.Tensor] = {}
        self._layout_ids: tp.Dict[int, int] = {}          # recording_index -> id of its layout content
        self._layout_bytes: tp.Dict[bytes, int] = {}

    def get_recording_layout(self, recording) -> torch.Tensor:
        index = recording.recording_index
        if index in self._cache:
            return self._cache[index]
        if getattr(recording, "layout", None) is not None:
            positions = torch.as_tensor(recording.layout, dtype=torch.float32)
        else:
            positions = self._layout_from_mne(recording.mne_info)
        self._cache[index] = positions
        return positions

    def _layout_from_mne(self, info) -> torch.Tensor:
        """2-D layout of the channels named in an ``mne.Info`` (what mne.find_layout knows about),
        min-max normalised per axis over the channels that were found; the rest stay INVALID."""
        import mne  # only needed with real recordings
        layout = mne.find_layout(info)
        slot_of = {name: k for k, name in enumerate(layout.names)}
        found = [(ch, slot_of[name.rsplit("-", 1)[0]]) for ch, name in enumerate(info.ch_names)
                 if name.rsplit("-", 1)[0] in slot_of]
        self._invalid_names.update(name.rsplit("-", 1)[0] for name in info.ch_names
                                   if name.rsplit("-", 1)[0] not in slot_of)
        positions = torch.full((len(info.ch_names), 2), self.INVALID)
        if found:
            channels, slots = zip(*found)
            xy = torch.as_tensor(layout.pos[list(slots), :2], dtype=torch.float64)
            lo, hi = xy.min(0).values, xy.max(0).values
            positions[list(channels)] = ((xy - lo) / (hi - lo)).float()
        return positions

    def get_positions(self, batch):
        """[B, C, 2] like the reference (common.py:225-233)."""
        meg = batch.meg
        B, C, T = meg.shape
        positions = torch.full((B, C, 2), self.INVALID, device=meg.device)
        for idx in range(len(batch)):
            rec_pos = self.get_recording_layout(batch._recordings[idx])
            positions[idx, :len(rec_pos)] = rec_pos.to(meg.device)
        return positions

    def get_unique_layouts(self, batch, n_channels: int, device):
        """-> (positions_u [U, C, 2] on device, layout_index [B] int64 on device).

        Two cache levels keep a shuffled training stream cheap: the stacked layouts of a SET of
        recordings (in first-seen order) stay on the device, so a new batch over known recordings costs
        one host loop over its segments and one 8*B-byte upload of the index; a batch that repeats a
        previous segment->recording assignment exactly also re-uses the uploaded index."""
        rec_ids = tuple(batch._recordings[i].recording_index for i in range(len(batch)))
        key = (rec_ids, n_channels, str(device))
        hit = self._device_cache.get(key)
        if hit is not None:
            return hit
        # recordings with byte-identical layouts (all sessions of one MEG system) share a slot
        slots: tp.Dict[int, int] = {}
        first: tp.List[int] = []
        index: tp.List[int] = []
        for idx, rec_id in enumerate(rec_ids):
            rec_key = self._layout_ids.get(rec_id)
            if rec_key is None:
                raw = self.get_recording_layout(batch._recordings[idx]).contiguous().numpy().tobytes()
                rec_key = self._layout_ids[rec_id] = self._layout_bytes.setdefault(raw, len(self._layout_bytes))
            slot = slots.get(rec_key)
            if slot is None:
                slot = slots[rec_key] = len(first)
                first.append(idx)
            index.append(slot)
        set_key = (tuple(slots), n_channels, str(device))
        positions_u = self._layout_set_cache.get(set_key)
        if positions_u is None:
            rows = []
            for idx in first:
                pos = torch.full((n_channels, 2), self.INVALID)
                rec_pos = self.get_recording_layout(batch._recordings[idx])
                pos[:len(rec_pos)] = rec_pos
                rows.append(pos)
            positions_u = torch.stack(rows).to(device)
            if len(self._layout_set_cache) >= 64:
                self._layout_set_cache.pop(next(iter(self._layout_set_cache)))
            self._layout_set_cache[set_key] = positions_u
        layout_index = torch.tensor(index, dtype=torch.int64)
        if torch.device(device).type == "cuda":
            # pinned + asynchronous: a blocking pageable upload would make the host wait for everything queued on the
            # stream (the previous step's backward pass) once per step with a fresh segment -> recording assignment
            layout_index = layout_index.pin_memory().to(device, non_blocking=True)
        else:
            layout_index = layout_index.to(device)
        if len(self._device_cache) >= 64:
            self._device_cache.pop(next(iter(self._device_cache)))
        self._device_cache[key] = (positions_u, layout_index)
        return positions_u, layout_index

    def is_invalid(self, positions):
        return (positions == self.INVALID).all(dim=-1)


class FourierEmb(nn.Module):
    """bm/models/common.py:239-271."""
    def __init__(self, dimension: int = 256, margin: float = 0.2):
        super().__init__()
        n_freqs = (dimension // 2)**0.5
        assert int(n_freqs ** 2 * 2) == dimension
        self.dimension = dimension
        self.margin = margin

    def forward(self, positions):
        return H.fourier_emb(positions.contiguous(), self.dimension, self.margin)


class ChannelDropout(nn.Module):
    """bm/models/common.py:273-310 (`simpleconv.dropout`): sensors without a position are zeroed; in training the
    sensors within ``dropout`` of a random centre are dropped and, with ``rescale``, every sensor is divided by its
    probability of being kept (estimated from 100 more random centres, like the reference).  Off the hot path (no
    grid of the paper sets it): GPU torch ops on the de-duplicated layouts."""

    N_TESTS = 100

    def __init__(self, dropout: float = 0.1, rescale: bool = True):
        super().__init__()
        self.dropout = dropout
        self.rescale = rescale
        self.position_getter = PositionGetter()
        self.ban_center_override: tp.Optional[torch.Tensor] = None     # test hook: every draw returns this centre

    def _centers(self, n: int, device) -> torch.Tensor:
        if self.ban_center_override is not None:
            return self.ban_center_override.to(device, torch.float32).view(1, 2).expand(n, 2)
        return torch.rand(n, 2, device=device)

    def forward(self, meg, batch):
        if not self.dropout:
            return meg
        B, C, T = meg.shape
        positions_u, layout_index = self.position_getter.get_unique_layouts(batch, C, meg.device)
        valid_u = (~self.position_getter.is_invalid(positions_u)).float()                 # [U, C]
        factor_u = valid_u
        if self.training:
            kept = ((positions_u - self._centers(1, meg.device)[0]).norm(dim=-1) > self.dropout).float()
            factor_u = factor_u * kept
            if self.rescale:
                centers = self._centers(self.N_TESTS, meg.device)                         # [N, 2]
                dist = (positions_u[None] - centers[:, None, None, :]).norm(dim=-1)       # [N, U, C]
                # (the reference accumulates kept / n_tests in a loop: the same sum in the same order)
                proba_kept = torch.zeros_like(valid_u)
                for n in range(self.N_TESTS):
                    proba_kept += (dist[n] > self.dropout).float() / self.N_TESTS
                factor_u = factor_u / (1e-8 + proba_kept)
        return meg * factor_u[layout_index][:, :, None]


class ChannelMerger(nn.Module):
    """bm/models/common.py:312-362; parameter ``heads`` [chout, pos_dim]."""
    def __init__(self, chout: int, pos_dim: int = 256,
                 dropout: float = 0, usage_penalty: float = 0.,
                 n_subjects: int = 200, per_subject: bool = False):
        super().__init__()
        assert pos_dim % 4 == 0
        self.position_getter = PositionGetter()
        self.per_subject = per_subject
        if self.per_subject:      # off the hot path (no grid of the paper sets it): one set of heads per subject
            self.heads = nn.Parameter(torch.randn(n_subjects, chout, pos_dim, requires_grad=True))
        else:
            self.heads = nn.Parameter(torch.randn(chout, pos_dim, requires_grad=True))
        self.heads.data /= pos_dim ** 0.5
        self.dropout = dropout
        self.embedding = FourierEmb(pos_dim)
        self.usage_penalty = usage_penalty
        self._penalty = torch.tensor(0.)
        # test hook: parity tests inject the ban centre instead of drawing torch.rand(2) on the
        # device generator (CPU and GPU RNG streams differ)
        self.ban_center_override: tp.Optional[torch.Tensor] = None

    @property
    def training_penalty(self):
        return self._penalty.to(next(self.parameters()).device)

    def layouts_and_ban(self, meg, batch):
        """(positions_u [U, C, 2], layout_index [B], ban_center | None, radius): the layout-dependent inputs of the
        attention, shared by ``forward`` and by SimpleConv's composed front end."""
        B, C, T = meg.shape
        positions_u, layout_index = self.position_getter.get_unique_layouts(batch, C, meg.device)
        ban_center, radius = None, 0.
        if self.training and self.usage_penalty > 0.:
            # bm/models/common.py:359-361: usage = weights.mean(dim=(0, 1)).sum() -- every (segment, output channel)
            # row of the softmax sums to 1, so the mean over rows sums to 1 over the sensors: the penalty is the
            # CONSTANT usage_penalty (its gradient is round-off noise in the reference, exactly zero here)
            self._penalty = torch.tensor(float(self.usage_penalty))
        if self.training and self.dropout:
            if self.ban_center_override is not None:
                ban_center = self.ban_center_override.to(meg.device, torch.float32)
            else:
                ban_center = torch.rand(2, device=meg.device)        # common.py:343
            radius = float(self.dropout)
        return positions_u, layout_index, ban_center, radius

    def forward(self, meg, batch):
        positions_u, layout_index, ban_center, radius = self.layouts_and_ban(meg, batch)
        if self.per_subject:
            # bm/models/common.py:348-351: heads gathered per segment.  Here: one attention map per (layout, subject)
            # pair that occurs in the batch (torch.unique: one small host round trip, off the hot path); autograd
            # scatters the per-pair head gradients back into the [n_subjects, chout, pos_dim] parameter
            n_subjects = self.heads.shape[0]
            # range-checked on the device like SubjectLayers (the reference's `heads.gather` raises, common.py:349): an
            # index >= n_subjects sets the index-error flag (raised at the Solver's next check) and is clamped to 0
            # instead of silently aliasing another (layout, subject) pair
            subjects = BF.H.index_i32(batch.subject_index.to(meg.device, torch.int64).contiguous(),
                                      n_subjects).to(torch.int64)
            pairs, inverse = torch.unique(layout_index * n_subjects + subjects, return_inverse=True)
            return BF.ChannelMergerFn.apply(meg, self.heads[pairs % n_subjects].contiguous(),
                                            positions_u[pairs // n_subjects].contiguous(), inverse.contiguous(),
                                            ban_center, radius)
        return BF.ChannelMergerFn.apply(meg, self.heads, positions_u, layout_index, ban_center,
                                        radius)


def pad_multiple(x: torch.Tensor, base: int):
    length = x.shape[-1]
    target = math.ceil(length / base) * base
    return torch.nn.functional.pad(x, (0, target - length))


class DualPathRNN(nn.Module):
    """``dual_path`` of SimpleConv (bm/models/common.py:154-180): ``4 * depth`` single-layer LSTMs over the time axis,
    each added to its input.  Off the hot path (no grid of the paper switches it on): the recurrences are torch's
    own GPU LSTM (the composite ATen implementation, not the vendor RNN library: deterministic and always present);
    module names and initialisation are the reference's, so its checkpoints load.

    What the reference computes, restated (``test_against_reference_golden[dual_path]`` pins it): every LSTM runs
    over the WHOLE zero-padded sequence (the reference folds the sequence into chunks of ``inner_length`` but feeds
    the unfolded one, common.py:166-171); after the even-numbered LSTMs the output's time steps are re-ordered as if
    it were the folded one (step ``i * n + j`` moves to ``j * inner_length + i``, n = chunks), and the sequence is
    reversed after every odd-numbered one."""

    def __init__(self, channels: int, depth: int, inner_length: int = 10):
        super().__init__()
        self.lstms = nn.ModuleList([nn.LSTM(channels, channels, 1) for _ in range(depth * 4)])
        self.inner_length = inner_length

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        length = x.shape[-1]
        seq = pad_multiple(x, self.inner_length).permute(2, 0, 1).contiguous()          # [L', B, C]
        steps, batch, channels = seq.shape
        chunks = steps // self.inner_length
        vendor_rnn = torch.backends.cudnn.enabled
        torch.backends.cudnn.enabled = False
        try:
            for k, lstm in enumerate(self.lstms):
                out = lstm(seq)[0]
                if k % 2 == 0:
                    out = out.view(self.inner_length, chunks, batch, channels).transpose(0, 1) \
                        .reshape(steps, batch, channels)
                seq = seq + out
                if k % 2 == 1:
                    seq = seq.flip(0)
        finally:
            torch.backends.cudnn.enabled = vendor_rnn
        return seq[:length].permute(1, 2, 0).contiguous()
