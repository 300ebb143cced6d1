"""Host-side mirror of ``bm/models/simpleconv.py``: ``SimpleConv`` with the reference's
constructor signature (bm/models/simpleconv.py:23-77), module names and ``state_dict`` layout, so
that ``bm/train.py:84-86`` can build it unchanged and reference checkpoints load into it; the
forward pass is a sequence of libbmhip kernels (see ``brainmagick_amd.functional``).

Options the paper's grids never use and that are outside the MI355X hot path raise
``NotImplementedError`` at construction time (listed in DESIGN.md).
"""
import random
import typing as tp

import torch
from torch import nn

from .common import (
    ConvSequence, ScaledEmbedding, SubjectLayers, ChannelMerger, make_activation)
from .. import functional as BF
from .. import hip_ops as H


class SimpleConv(nn.Module):
    def __init__(self,
                 # Channels
                 in_channels: tp.Dict[str, int],
                 out_channels: int,
                 hidden: tp.Dict[str, int],
                 # Overall structure
                 depth: int = 4,
                 concatenate: bool = False,  # concatenate the inputs
                 linear_out: bool = False,
                 complex_out: bool = False,
                 # Conv layer
                 kernel_size: int = 5,
                 growth: float = 1.,
                 dilation_growth: int = 2,
                 dilation_period: tp.Optional[int] = None,
                 skip: bool = False,
                 post_skip: bool = False,
                 scale: tp.Optional[float] = None,
                 rewrite: bool = False,
                 groups: int = 1,
                 glu: int = 0,
                 glu_context: int = 0,
                 glu_glu: bool = True,
                 gelu: bool = False,
                 # Dual path RNN
                 dual_path: int = 0,
                 # Dropouts, BN, activations
                 conv_dropout: float = 0.0,
                 dropout_input: float = 0.0,
                 batch_norm: bool = False,
                 relu_leakiness: float = 0.0,
                 # Subject specific settings
                 n_subjects: int = 200,
                 subject_dim: int = 64,
                 subject_layers: bool = False,
                 subject_layers_dim: str = "input",  # or hidden
                 subject_layers_id: bool = False,
                 embedding_scale: float = 1.0,
                 # stft transform
                 n_fft: tp.Optional[int] = None,
                 fft_complex: bool = True,
                 # Attention multi-dataset support
                 merger: bool = False,
                 merger_pos_dim: int = 256,
                 merger_channels: int = 270,
                 merger_dropout: float = 0.2,
                 merger_penalty: float = 0.,
                 merger_per_subject: bool = False,
                 dropout: float = 0.,
                 dropout_rescale: bool = True,
                 initial_linear: int = 0,
                 initial_depth: int = 1,
                 initial_nonlin: bool = False,
                 subsample_meg_channels: int = 0,
                 ):
        super().__init__()
        if set(in_channels.keys()) != set(hidden.keys()):
            raise ValueError("Channels and hidden keys must match "
                             f"({set(in_channels.keys())} and {set(hidden.keys())})")
        off_path = dict(concatenate=concatenate, dual_path=bool(dual_path), n_fft=n_fft is not None,
                        dropout=dropout > 0., multi_input=set(in_channels) != {"meg"})
        bad = [k for k, v in off_path.items() if v]
        if bad:
            raise NotImplementedError(
                f"SimpleConv options {bad} are outside the MI355X hot path (STFT / DualPathRNN / "
                "ChannelDropout / multi-input are unused by the paper's grids, SURVEY.md §2.2)")
        self._concatenate = concatenate
        self.out_channels = out_channels
        activation = make_activation(gelu, relu_leakiness)
        assert kernel_size % 2 == 1, "For padding to work, this must be verified"

        self.merger = None
        self.dropout = None
        self.subsampled_meg_channels: tp.Optional[list] = None
        if subsample_meg_channels:
            assert 'meg' in in_channels
            indexes = list(range(in_channels['meg']))
            rng = random.Random(1234)
            rng.shuffle(indexes)
            self.subsampled_meg_channels = indexes[:subsample_meg_channels]
            keep = torch.zeros(1, in_channels['meg'], 1)
            keep[:, self.subsampled_meg_channels] = 1.
            self.register_buffer("_channel_keep", keep, persistent=False)

        self.initial_linear = None
        if merger:
            self.merger = ChannelMerger(
                merger_channels, pos_dim=merger_pos_dim, dropout=merger_dropout,
                usage_penalty=merger_penalty, n_subjects=n_subjects, per_subject=merger_per_subject)
            in_channels["meg"] = merger_channels

        if initial_linear:
            init = [nn.Conv1d(in_channels["meg"], initial_linear, 1)]
            for _ in range(initial_depth - 1):
                init += [activation(), nn.Conv1d(initial_linear, initial_linear, 1)]
            if initial_nonlin:
                init += [activation()]
            self.initial_linear = nn.Sequential(*init)
            in_channels["meg"] = initial_linear

        self.subject_layers = None
        if subject_layers:
            assert "meg" in in_channels
            meg_dim = in_channels["meg"]
            dim = {"hidden": hidden["meg"], "input": meg_dim}[subject_layers_dim]
            self.subject_layers = SubjectLayers(meg_dim, dim, n_subjects, subject_layers_id)
            in_channels["meg"] = dim

        self.stft = None
        self.subject_embedding = None
        if subject_dim:
            self.subject_embedding = ScaledEmbedding(n_subjects, subject_dim, embedding_scale)
            in_channels["meg"] += subject_dim

        # sequence of channel sizes of the conv stack
        sizes = {}
        for name in in_channels:
            sizes[name] = [in_channels[name]]
            sizes[name] += [int(round(hidden[name] * growth ** k)) for k in range(depth)]

        params: tp.Dict[str, tp.Any]
        params = dict(kernel=kernel_size, stride=1,
                      leakiness=relu_leakiness, dropout=conv_dropout, dropout_input=dropout_input,
                      batch_norm=batch_norm, dilation_growth=dilation_growth, groups=groups,
                      dilation_period=dilation_period, skip=skip, post_skip=post_skip, scale=scale,
                      rewrite=rewrite, glu=glu, glu_context=glu_context, glu_glu=glu_glu,
                      activation=activation)

        final_channels = sum([x[-1] for x in sizes.values()])
        self.dual_path = None
        self.final = None
        if linear_out:
            assert not complex_out
            self.final = nn.ConvTranspose1d(final_channels, out_channels, 1, 1, 0)
        elif complex_out:
            self.final = nn.Sequential(
                nn.Conv1d(final_channels, 2 * final_channels, 1),
                activation(),
                nn.ConvTranspose1d(2 * final_channels, out_channels, 1, 1, 0))
        else:
            assert len(sizes) == 1, "if no linear_out, there must be a single branch."
            params['activation_on_last'] = False
            list(sizes.values())[0][-1] = out_channels

        self.encoders = nn.ModuleDict({name: ConvSequence(channels, **params)
                                       for name, channels in sizes.items()})

    def forward(self, inputs, batch):
        subjects = batch.subject_index
        length = next(iter(inputs.values())).shape[-1]  # length of any of the inputs
        x = inputs["meg"]
        if not x.is_cuda:
            raise RuntimeError("brainmagick_amd.SimpleConv runs on the MI355X HIP path only; got a "
                               f"{x.device} tensor (there is no CPU fallback)")
        if x.dtype != torch.float32:
            raise TypeError(f"SimpleConv expects fp32 inputs like the reference, got {x.dtype}")

        if self.subsampled_meg_channels is not None:
            x = x * self._channel_keep                       # constant 0/1 mask, simpleconv.py:202-205
        if self.merger is not None:
            x = self.merger(x, batch)
        if self.initial_linear is not None:
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
        if self.subject_layers is not None:
            x = self.subject_layers(x, subjects)
        if self.subject_embedding is not None:
            emb = self.subject_embedding(subjects)[:, :, None]
            x = torch.cat([x, emb.expand(-1, -1, length)], dim=1)
        inputs["meg"] = x                                     # the reference reassigns the dict entry

        x = self.encoders["meg"](x)
        if self.final is not None:
            if isinstance(self.final, nn.ConvTranspose1d):
                x = BF.Conv1dFn.apply(x, self.final.weight, self.final.bias, 1, H.ACT_NONE, 0., True)
            else:
                conv, act, tconv = self.final[0], self.final[1], self.final[2]
                x = BF.Conv1dFn.apply(x, conv.weight, conv.bias, 1, act.code, act.leak, False)
                x = BF.Conv1dFn.apply(x, tconv.weight, tconv.bias, 1, H.ACT_NONE, 0., True)
        assert x.shape[-1] >= length
        return x[:, :, :length]
