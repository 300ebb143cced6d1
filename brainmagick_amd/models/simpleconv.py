"""Host-side mirror of ``bm/models/simpleconv.py``: ``SimpleConv`` with the reference's
constructor signature (bm/models/simpleconv.py:23-77), module names and ``state_dict`` layout, so
that ``bm/train.py:84-86`` can build it unchanged and reference checkpoints load into it; the
forward pass is a sequence of libbmhip kernels (see ``brainmagick_amd.functional``).

Options the paper's grids never use are either implemented off the hot path (GPU torch ops around the fused layer
functions: ``dropout``, ``conv_dropout``, ``dropout_input``, ``scale``, ``rewrite``, ``post_skip``, ``merger_penalty``,
``merger_per_subject``, ``groups``, more inputs than ``meg`` with or without ``concatenate``) or raise
``NotImplementedError`` at construction time (``n_fft``).  ``dual_path`` runs torch's GPU LSTM
between the stack and the head.
"""
import random
import typing as tp

import torch
from torch import nn

from .common import (
    ConvSequence, ScaledEmbedding, SubjectLayers, ChannelMerger, ChannelDropout, DualPathRNN, make_activation)
from .. import functional as BF
from .. import hip_ops as H


# A/B switch (measurement / tests): "0" keeps the three front-end layers as three kernels even where they compose
_FUSE_FRONT_END = True      # tests flip it to compare the composed front end with the three-layer chain


class SimpleConv(nn.Module):
    """Constructor keywords, defaults, sub-module names and ``state_dict`` keys are the reference's
    (bm/models/simpleconv.py:23-77): ``bm/train.py:84-86`` passes ``**args.simpleconv`` and reference
    checkpoints load with ``strict=True``.  Keywords are grouped here by what this implementation does with
    them: built on the HIP path / accepted for signature compatibility / rejected."""

    def __init__(self, in_channels: tp.Dict[str, int], out_channels: int, hidden: tp.Dict[str, int],
                 depth: int = 4, concatenate: bool = False, linear_out: bool = False,
                 complex_out: bool = False, kernel_size: int = 5, growth: float = 1.,
                 dilation_growth: int = 2, dilation_period: tp.Optional[int] = None, skip: bool = False,
                 post_skip: bool = False, scale: tp.Optional[float] = None, rewrite: bool = False,
                 groups: int = 1, glu: int = 0, glu_context: int = 0, glu_glu: bool = True,
                 gelu: bool = False, dual_path: int = 0, conv_dropout: float = 0.0,
                 dropout_input: float = 0.0, batch_norm: bool = False, relu_leakiness: float = 0.0,
                 n_subjects: int = 200, subject_dim: int = 64, subject_layers: bool = False,
                 subject_layers_dim: str = "input", subject_layers_id: bool = False,
                 embedding_scale: float = 1.0, n_fft: tp.Optional[int] = None, fft_complex: bool = True,
                 merger: bool = False, merger_pos_dim: int = 256, merger_channels: int = 270,
                 merger_dropout: float = 0.2, merger_penalty: float = 0., merger_per_subject: bool = False,
                 dropout: float = 0., dropout_rescale: bool = True, initial_linear: int = 0,
                 initial_depth: int = 1, initial_nonlin: bool = False, subsample_meg_channels: int = 0):
        super().__init__()
        if set(in_channels) != set(hidden):
            raise ValueError("Channels and hidden keys must match "
                             f"({set(in_channels.keys())} and {set(hidden.keys())})")
        if n_fft is not None:
            raise NotImplementedError(
                "SimpleConv option n_fft (torchaudio STFT front end) is outside the MI355X hot path: no grid of the "
                "paper uses it (SURVEY.md §2.2), and torchaudio being absent from this image the reference itself "
                "cannot run it here, so there would be nothing to hold an implementation to")
        if "meg" not in in_channels:
            raise ValueError("SimpleConv needs a 'meg' input (the sensor front end is built around it)")
        if kernel_size % 2 != 1:
            raise AssertionError("For padding to work, this must be verified")       # reference message
        if linear_out and complex_out:
            raise AssertionError("linear_out and complex_out are exclusive")
        self.out_channels = out_channels
        self._concatenate = concatenate
        activation = make_activation(gelu, relu_leakiness)
        in_channels, hidden = dict(in_channels), dict(hidden)      # (the reference edits its caller's dict)

        width = in_channels["meg"]
        self.dropout = ChannelDropout(dropout, dropout_rescale) if dropout > 0. else None   # simpleconv.py:103-104
        self.subsampled_meg_channels: tp.Optional[list] = None
        if subsample_meg_channels:
            self._subsample(width, subsample_meg_channels)
        width = self._build_front_end(
            width, activation, hidden["meg"], n_subjects,
            merger=dict(on=merger, channels=merger_channels, pos_dim=merger_pos_dim, dropout=merger_dropout,
                        penalty=merger_penalty, per_subject=merger_per_subject),
            linear=dict(width=initial_linear, depth=initial_depth, nonlin=initial_nonlin),
            subject=dict(layers=subject_layers, where=subject_layers_dim, init_id=subject_layers_id,
                         emb_dim=subject_dim, emb_scale=embedding_scale))

        # one conv stack per input (or one over the channel-wise concatenation of all inputs, simpleconv.py:138-147)
        # and the head; without a head the last conv of the single stack produces the output channels
        in_channels["meg"] = width
        if concatenate:
            in_channels = {"concat": sum(in_channels.values())}
            hidden = {"concat": sum(hidden.values())}
        widths = {name: [in_channels[name]] + [int(round(hidden[name] * growth ** k)) for k in range(depth)]
                  for name in in_channels}
        stack_kw = dict(kernel=kernel_size, stride=1, leakiness=relu_leakiness, dropout=conv_dropout,
                        dropout_input=dropout_input, batch_norm=batch_norm, dilation_growth=dilation_growth,
                        groups=groups, dilation_period=dilation_period, skip=skip, post_skip=post_skip,
                        scale=scale, rewrite=rewrite, glu=glu, glu_context=glu_context, glu_glu=glu_glu,
                        activation=activation)
        top = sum(w[-1] for w in widths.values())
        self.dual_path = DualPathRNN(top, dual_path) if dual_path else None       # simpleconv.py:163-165
        if linear_out:
            self.final = nn.ConvTranspose1d(top, out_channels, 1, 1, 0)
        elif complex_out:
            self.final = nn.Sequential(nn.Conv1d(top, 2 * top, 1), activation(),
                                       nn.ConvTranspose1d(2 * top, out_channels, 1, 1, 0))
        else:
            assert len(widths) == 1, "if no linear_out, there must be a single branch."      # reference message
            self.final = None
            stack_kw["activation_on_last"] = False
            next(iter(widths.values()))[-1] = out_channels
        self.encoders = nn.ModuleDict({name: ConvSequence(w, **stack_kw) for name, w in widths.items()})

    def _subsample(self, n_sensors: int, keep_n: int):
        """`subsample_meg_channels`: a fixed pseudo-random subset of sensors (seed 1234, simpleconv.py:95-100)
        survives, as a constant 0/1 mask applied on the device."""
        order = list(range(n_sensors))
        random.Random(1234).shuffle(order)
        self.subsampled_meg_channels = order[:keep_n]
        keep = torch.zeros(1, n_sensors, 1)
        keep[:, self.subsampled_meg_channels] = 1.
        self.register_buffer("_channel_keep", keep, persistent=False)

    def _build_front_end(self, width: int, activation, hidden: int, n_subjects: int, merger: dict,
                         linear: dict, subject: dict) -> int:
        """Spatial attention -> 1x1 convs -> per-subject linear map -> subject embedding
        (simpleconv.py:102-135); returns the channel count handed to the conv stack."""
        self.merger = self.initial_linear = self.subject_layers = self.subject_embedding = None
        if merger["on"]:
            self.merger = ChannelMerger(merger["channels"], pos_dim=merger["pos_dim"], dropout=merger["dropout"],
                                        usage_penalty=merger["penalty"], n_subjects=n_subjects,
                                        per_subject=merger["per_subject"])
            width = merger["channels"]
        if linear["width"]:
            layers: tp.List[nn.Module] = [nn.Conv1d(width, linear["width"], 1)]
            for _ in range(linear["depth"] - 1):
                layers += [activation(), nn.Conv1d(linear["width"], linear["width"], 1)]
            if linear["nonlin"]:
                layers.append(activation())
            self.initial_linear = nn.Sequential(*layers)
            width = linear["width"]
        if subject["layers"]:
            out = {"hidden": hidden, "input": width}[subject["where"]]
            self.subject_layers = SubjectLayers(width, out, n_subjects, subject["init_id"])
            width = out
        if subject["emb_dim"]:
            self.subject_embedding = ScaledEmbedding(n_subjects, subject["emb_dim"], subject["emb_scale"])
            width += subject["emb_dim"]
        return width

    def _front_end_is_linear_chain(self) -> bool:
        """merger -> ONE 1x1 conv (no activation) -> subject layers: the paper's front end
        (conf/model/clip_conv.yaml: initial_depth 1, no initial_nonlin), composable into one grouped 1x1 conv."""
        return (self.merger is not None and self.subject_layers is not None and self.initial_linear is not None
                and len(self.initial_linear) == 1 and self.subsampled_meg_channels is None
                and not self.merger.per_subject)

    def forward(self, inputs, batch):
        subjects = batch.subject_index
        length = next(iter(inputs.values())).shape[-1]
        x = inputs["meg"]
        if not x.is_cuda:
            raise RuntimeError("brainmagick_amd.SimpleConv runs on the MI355X HIP path only; got a "
                               f"{x.device} tensor (there is no CPU fallback)")
        if x.dtype != torch.float32:
            raise TypeError(f"SimpleConv expects fp32 inputs like the reference, got {x.dtype}")

        if self.subsampled_meg_channels is not None:
            x = x * self._channel_keep                       # constant 0/1 mask, simpleconv.py:202-205
        if self.dropout is not None:
            x = self.dropout(x, batch)                       # simpleconv.py:207-208
        fused = self._front_end_is_linear_chain() and _FUSE_FRONT_END
        if fused:
            positions_u, layout_index, ban_center, radius = self.merger.layouts_and_ban(x, batch)
            # one composed matrix per (layout, subject) pair: worth it while there are fewer pairs than segments
            fused = positions_u.shape[0] * self.subject_layers.weights.shape[0] <= x.shape[0]
        self.front_end_fused = fused              # introspection (tests, bench)
        if fused:
            conv = self.initial_linear[0]
            x = BF.FusedFrontEndFn.apply(x, self.merger.heads, conv.weight, conv.bias, self.subject_layers.weights,
                                         positions_u, layout_index, subjects, ban_center, radius)
        elif self.merger is not None and self._front_end_is_linear_chain() and _FUSE_FRONT_END:
            # the layouts (and the random ban centre) were drawn above: do not draw them twice
            x = BF.ChannelMergerFn.apply(x, self.merger.heads, positions_u, layout_index, ban_center, radius)
        elif self.merger is not None:
            x = self.merger(x, batch)
        if fused:
            pass
        elif self.initial_linear is not None:
            mods = list(self.initial_linear)
            i = 0
            while i < len(mods):
                conv = mods[i]
                act = mods[i + 1] if i + 1 < len(mods) and not isinstance(mods[i + 1], nn.Conv1d) \
                    else None
                x = BF.Conv1dFn.apply(x, conv.weight, conv.bias, 1,
                                      act.code if act is not None else H.ACT_NONE,
                                      act.leak if act is not None else 0., False)
                i += 2 if act is not None else 1
        if self.subject_layers is not None and not fused:
            x = self.subject_layers(x, subjects)
        if self.subject_embedding is not None:
            emb = self.subject_embedding(subjects)[:, :, None]
            x = torch.cat([x, emb.expand(-1, -1, length)], dim=1)
        inputs["meg"] = x                                     # the reference reassigns the dict entry

        if self._concatenate:                                 # inputs side by side, in the order of their names
            x = self.encoders["concat"](torch.cat([v for _, v in sorted(inputs.items())], dim=1))
        elif len(self.encoders) == 1:
            x = self.encoders["meg"](x)
        else:
            x = torch.cat([self.encoders[name](v) for name, v in sorted(inputs.items())], dim=1)
        if self.dual_path is not None:
            x = self.dual_path(x)
        if self.final is not None:
            if isinstance(self.final, nn.ConvTranspose1d):
                x = BF.Conv1dFn.apply(x, self.final.weight, self.final.bias, 1, H.ACT_NONE, 0., True)
            else:
                conv, act, tconv = self.final[0], self.final[1], self.final[2]
                x = BF.Conv1dFn.apply(x, conv.weight, conv.bias, 1, act.code, act.leak, False)
                x = BF.Conv1dFn.apply(x, tconv.weight, tconv.bias, 1, H.ACT_NONE, 0., True)
        assert x.shape[-1] >= length
        # (the slice is the reference's; when it is the identity the tensor itself goes on, with the maximum its
        # producer published for the ClipLoss contraction)
        return x if x.shape[-1] == length else x[:, :, :length]
