"""CPU oracle for the brainmagick SimpleConv + ClipLoss hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32 or fp64) *functional restatement* of the reference's
algorithm for the path named in BASELINE.json.  Every function cites the reference file:line it
follows (paths relative to the upstream repo root).  It is driven by the reference's own
``state_dict`` keys so that a reference checkpoint, this oracle and the HIP product all consume
the same tensors.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg -- as the checker / reported baseline only.  The product (``brainmagick_amd``) never imports
it and has no CPU fallback.

Parity pin: the reference has NO golden vectors or known-answer tests for this path
(SURVEY.md §8c), so the oracle is pinned against outputs of the *live reference code* imported
in the build container: ``tests/golden/make_golden.py`` runs the real ``bm.models.simpleconv``
/ ``bm.losses`` on seeded inputs and commits inputs+outputs+grads as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this file against them.

Third-party arithmetic: PyTorch ops (conv1d / batch_norm / gelu / cross_entropy / Adam) are the
engine of the reference itself (requirements.txt: torch>=1.5, unpinned; here 2.10.0) and of this
oracle.  ``flashy.distrib.sync_model`` (absent) is restated in ``sync_gradients_reference`` from
its call site bm/solver.py:386 -- parity for that function is "unpinned" (no source, no test).
"""
import math
import typing as tp

import torch
from torch.nn import functional as F

INVALID = -0.1  # bm/models/common.py:188 PositionGetter.INVALID

# Defaults of SimpleConv.__init__ (bm/models/simpleconv.py:23-77) for the kwargs this path honours.
SIMPLECONV_DEFAULTS: tp.Dict[str, tp.Any] = dict(
    depth=4, concatenate=False, linear_out=False, complex_out=False, kernel_size=5, growth=1.,
    dilation_growth=2, dilation_period=None, skip=False, post_skip=False, scale=None,
    rewrite=False, groups=1, glu=0, glu_context=0, glu_glu=True, gelu=False, dual_path=0,
    conv_dropout=0.0, dropout_input=0.0, batch_norm=False, relu_leakiness=0.0, n_subjects=200,
    subject_dim=64, subject_layers=False, subject_layers_dim="input", subject_layers_id=False,
    embedding_scale=1.0, n_fft=None, fft_complex=True, merger=False, merger_pos_dim=256,
    merger_channels=270, merger_dropout=0.2, merger_penalty=0., merger_per_subject=False,
    dropout=0., dropout_rescale=True, initial_linear=0, initial_depth=1, initial_nonlin=False,
    subsample_meg_channels=0)

# conf/model/clip_conv.yaml:5-38 over conf/model_defaults/defaults.yaml:35-82 (the paper model).
CLIP_CONV_CFG: tp.Dict[str, tp.Any] = dict(
    depth=10, kernel_size=3, dilation_growth=2, dilation_period=5, batch_norm=True, skip=True,
    gelu=True, glu=2, glu_context=1, glu_glu=True, complex_out=True, merger=True,
    merger_pos_dim=2048, merger_channels=270, merger_dropout=0.2, merger_penalty=0.,
    initial_linear=270, initial_depth=1, subject_layers=True, subject_layers_dim="input",
    subject_dim=0)


def full_cfg(cfg: tp.Dict[str, tp.Any]) -> tp.Dict[str, tp.Any]:
    out = dict(SIMPLECONV_DEFAULTS)
    out.update(cfg)
    return out


# ----------------------------------------------------------------------------------------------
# ChannelMerger front end
# ----------------------------------------------------------------------------------------------
def fourier_emb(positions: torch.Tensor, dimension: int, margin: float = 0.2) -> torch.Tensor:
    """bm/models/common.py:254-271 FourierEmb.forward.  positions [..., 2] -> [..., dimension]."""
    *O, D = positions.shape
    assert D == 2
    n_freqs = (dimension // 2) ** 0.5
    assert int(n_freqs ** 2 * 2) == dimension
    freqs_y = torch.arange(n_freqs).to(positions)
    freqs_x = freqs_y[:, None]
    width = 1 + 2 * margin
    positions = positions + margin
    p_x = 2 * math.pi * freqs_x / width
    p_y = 2 * math.pi * freqs_y / width
    positions = positions[..., None, None, :]
    loc = (positions[..., 0] * p_x + positions[..., 1] * p_y).view(*O, -1)
    return torch.cat([torch.cos(loc), torch.sin(loc)], dim=-1)


def is_invalid(positions: torch.Tensor) -> torch.Tensor:
    """bm/models/common.py:235-236.  Positions are produced in fp32 (common.py:214); the
    comparison is done in fp32 so that an fp64 oracle run sees the same mask."""
    return (positions.to(torch.float32) == INVALID).all(dim=-1)


def merger_weights(heads: torch.Tensor, positions: torch.Tensor, training: bool, dropout: float,
                   ban_center: tp.Optional[torch.Tensor]) -> torch.Tensor:
    """bm/models/common.py:337-357: masked softmax-attention weights over sensors, [B, O, C].

    ``ban_center`` replaces the ``torch.rand(2)`` of common.py:343 (device RNG streams differ
    between CPU and GPU, so parity tests inject it)."""
    B, C, _ = positions.shape
    embedding = fourier_emb(positions, heads.shape[-1])
    score_offset = torch.zeros(B, C, dtype=positions.dtype)
    score_offset[is_invalid(positions)] = float('-inf')
    if training and dropout:
        assert ban_center is not None, "training with merger_dropout needs an injected ban centre"
        banned = (positions - ban_center.to(positions)).norm(dim=-1) <= dropout
        score_offset[banned] = float('-inf')
    heads_b = heads[None].expand(B, -1, -1)
    scores = torch.einsum("bcd,bod->boc", embedding, heads_b)
    scores = scores + score_offset[:, None]
    return torch.softmax(scores, dim=2)


def channel_merger(meg: torch.Tensor, heads: torch.Tensor, positions: torch.Tensor,
                   training: bool, dropout: float, ban_center=None) -> torch.Tensor:
    """bm/models/common.py:334-362 ChannelMerger.forward (per_subject=False, usage_penalty=0)."""
    weights = merger_weights(heads, positions, training, dropout, ban_center)
    return torch.einsum("bct,boc->bot", meg, weights)


def subject_layers(x: torch.Tensor, weights: torch.Tensor, subjects: torch.Tensor) -> torch.Tensor:
    """bm/models/common.py:55-58 SubjectLayers.forward."""
    _, C, D = weights.shape
    w = weights.gather(0, subjects.view(-1, 1, 1).expand(-1, C, D))
    return torch.einsum("bct,bcd->bdt", x, w)


# ----------------------------------------------------------------------------------------------
# ConvSequence
# ----------------------------------------------------------------------------------------------
def _activation(x, cfg):
    """bm/models/simpleconv.py:85-90: GELU (exact erf) | LeakyReLU(leak) | ReLU."""
    if cfg["gelu"]:
        return F.gelu(x)
    if cfg["relu_leakiness"]:
        return F.leaky_relu(x, cfg["relu_leakiness"])
    return F.relu(x)


def conv_sequence_plan(channels: tp.Sequence[int], cfg) -> tp.List[dict]:
    """Static structure of ConvSequence.__init__ (bm/models/common.py:81-140): per layer
    dilation / padding / whether BN+activation / whether a GLU block follows."""
    plan = []
    dilation = 1
    kernel = cfg["kernel_size"]
    activation_on_last = cfg.get("_activation_on_last", True)
    n = len(channels) - 1
    for k, (chin, chout) in enumerate(zip(channels[:-1], channels[1:])):
        is_last = k == n - 1
        if cfg["dilation_period"] and (k % cfg["dilation_period"]) == 0:
            dilation = 1
        pad = kernel // 2 * dilation
        layer = dict(chin=chin, chout=chout, dilation=dilation, pad=pad,
                     act=activation_on_last or not is_last,
                     glu=bool(cfg["glu"] and (k + 1) % cfg["glu"] == 0))
        plan.append(layer)
        dilation *= cfg["dilation_growth"]
    return plan


def conv_sequence(x: torch.Tensor, sd: dict, prefix: str, channels, cfg, training: bool,
                  new_buffers: tp.Optional[dict] = None) -> torch.Tensor:
    """bm/models/common.py:142-151 ConvSequence.forward with the module list built at :96-140.

    BatchNorm1d semantics = torch defaults (eps 1e-5, momentum 0.1): training normalises with the
    biased batch variance over (B,T) and updates running_var with the unbiased one."""
    assert cfg["groups"] == 1 and not cfg["rewrite"] and not cfg["post_skip"] \
        and cfg["scale"] is None and not cfg["conv_dropout"] and not cfg["dropout_input"]
    dot = f"{prefix}." if prefix else ""
    for k, layer in enumerate(conv_sequence_plan(channels, cfg)):
        old_x = x
        p = f"{dot}sequence.{k}"
        x = F.conv1d(x, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"], stride=1,
                     padding=layer["pad"], dilation=layer["dilation"])
        if layer["act"]:
            if cfg["batch_norm"]:
                rm, rv = sd[f"{p}.1.running_mean"], sd[f"{p}.1.running_var"]
                if training:
                    rm, rv = rm.clone(), rv.clone()
                x = F.batch_norm(x, rm, rv, sd[f"{p}.1.weight"], sd[f"{p}.1.bias"],
                                 training=training, momentum=0.1, eps=1e-5)
                if training and new_buffers is not None:
                    new_buffers[f"{p}.1.running_mean"] = rm
                    new_buffers[f"{p}.1.running_var"] = rv
                    new_buffers[f"{p}.1.num_batches_tracked"] = \
                        sd[f"{p}.1.num_batches_tracked"] + 1
            x = _activation(x, cfg)
        if cfg["skip"] and x.shape == old_x.shape:
            x = x + old_x
        if layer["glu"]:
            g = f"{dot}glus.{k}.0"
            x = F.conv1d(x, sd[f"{g}.weight"], sd[f"{g}.bias"], padding=cfg["glu_context"])
            x = F.glu(x, dim=1) if cfg["glu_glu"] else _activation(x, cfg)
    return x


# ----------------------------------------------------------------------------------------------
# SimpleConv
# ----------------------------------------------------------------------------------------------
def simpleconv_channels(in_channels: int, out_channels: int, hidden: int, cfg) -> tp.List[int]:
    """Channel bookkeeping of SimpleConv.__init__ (bm/models/simpleconv.py:104-196)."""
    cfg = full_cfg(cfg)
    c = in_channels
    if cfg["merger"]:
        c = cfg["merger_channels"]
    if cfg["initial_linear"]:
        c = cfg["initial_linear"]
    if cfg["subject_layers"]:
        c = {"hidden": hidden, "input": c}[cfg["subject_layers_dim"]]
    if cfg["subject_dim"]:
        c += cfg["subject_dim"]
    sizes = [c] + [int(round(hidden * cfg["growth"] ** k)) for k in range(cfg["depth"])]
    if not cfg["linear_out"] and not cfg["complex_out"]:
        sizes[-1] = out_channels
    return sizes


def subsampled_channels(n_channels: int, n_keep: int) -> tp.List[int]:
    """bm/models/simpleconv.py:97-102: fixed pseudo-random channel subset, Random(1234)."""
    import random
    indexes = list(range(n_channels))
    rng = random.Random(1234)
    rng.shuffle(indexes)
    return indexes[:n_keep]


def simpleconv_forward(sd: dict, cfg: dict, meg: torch.Tensor, positions: torch.Tensor,
                       subjects: torch.Tensor, hidden: int, out_channels: int,
                       training: bool = False, ban_center=None,
                       new_buffers: tp.Optional[dict] = None) -> torch.Tensor:
    """bm/models/simpleconv.py:198-249 SimpleConv.forward for inputs={'meg': meg}.

    ``positions`` [B,C,2] is what PositionGetter.get_positions (common.py:225-233) returns."""
    cfg = full_cfg(cfg)
    assert not cfg["concatenate"] and cfg["n_fft"] is None and not cfg["dual_path"] \
        and not cfg["dropout"] and not cfg["merger_per_subject"] and not cfg["merger_penalty"]
    in_channels = meg.shape[1]
    length = meg.shape[-1]
    x = meg
    if cfg["subsample_meg_channels"]:
        mask = torch.zeros_like(x[:1, :, :1])
        mask[:, subsampled_channels(in_channels, cfg["subsample_meg_channels"])] = 1.
        x = x * mask
    if cfg["merger"]:
        x = channel_merger(x, sd["merger.heads"], positions, training, cfg["merger_dropout"],
                           ban_center)
    if cfg["initial_linear"]:
        # simpleconv.py:113-120: Conv1d(k=1) [+ act + Conv1d]*(depth-1) [+ act]
        idx = 0
        x = F.conv1d(x, sd["initial_linear.0.weight"], sd["initial_linear.0.bias"])
        for _ in range(cfg["initial_depth"] - 1):
            x = _activation(x, cfg)
            idx += 2
            x = F.conv1d(x, sd[f"initial_linear.{idx}.weight"], sd[f"initial_linear.{idx}.bias"])
        if cfg["initial_nonlin"]:
            x = _activation(x, cfg)
    if cfg["subject_layers"]:
        x = subject_layers(x, sd["subject_layers.weights"], subjects)
    if cfg["subject_dim"]:
        # simpleconv.py:230-232 with ScaledEmbedding (common.py:29-43): embedding(x) * scale
        emb = F.embedding(subjects, sd["subject_embedding.embedding.weight"]) \
            * cfg["embedding_scale"]
        x = torch.cat([x, emb[:, :, None].expand(-1, -1, length)], dim=1)
    sizes = simpleconv_channels(in_channels, out_channels, hidden, cfg)
    seq_cfg = dict(cfg)
    seq_cfg["_activation_on_last"] = bool(cfg["linear_out"] or cfg["complex_out"])
    x = conv_sequence(x, sd, "encoders.meg", sizes, seq_cfg, training, new_buffers)
    if cfg["linear_out"]:
        # ConvTranspose1d(k=1,s=1,p=0): weight stored (in, out, 1)
        x = F.conv_transpose1d(x, sd["final.weight"], sd["final.bias"])
    elif cfg["complex_out"]:
        x = F.conv1d(x, sd["final.0.weight"], sd["final.0.bias"])
        x = _activation(x, cfg)
        x = F.conv_transpose1d(x, sd["final.2.weight"], sd["final.2.bias"])
    assert x.shape[-1] >= length
    return x[:, :, :length]


# ----------------------------------------------------------------------------------------------
# ClipLoss
# ----------------------------------------------------------------------------------------------
DEEP_MEL_CFG: tp.Dict[str, tp.Any] = dict(      # conf/feature_model/deep_mel.yaml
    kernel_size=3, dilation_growth=2, dilation_period=5, batch_norm=True, skip=True, glu=2,
    glu_context=1, glu_glu=True, gelu=False, relu_leakiness=0.0, groups=1, rewrite=False,
    post_skip=False, scale=None, conv_dropout=0.0, dropout_input=0.0, _activation_on_last=False)


def deep_mel_forward(sd: dict, features: torch.Tensor, n_hidden_channels: int, n_hidden_layers: int,
                     n_out_channels: int, training: bool, cfg=None,
                     new_buffers: tp.Optional[dict] = None) -> torch.Tensor:
    """bm/models/features.py:15-35 DeepMel = ConvSequence([F] + [hidden]*(L-1) + [out]) with the
    default activation LeakyReLU(0.0) (bm/models/common.py:97-98)."""
    cfg = dict(DEEP_MEL_CFG if cfg is None else cfg)
    channels = [features.shape[1]] + [n_hidden_channels] * (n_hidden_layers - 1) + [n_out_channels]
    return conv_sequence(features, sd, "", channels, cfg, training, new_buffers)


def clip_trim(estimates, candidates, tmin=None, tmax=None, dset_tmin=None, sample_rate=None):
    """bm/losses.py:50-75 ClipLoss.trim_samples (tmin/tmax already resolved for train/eval)."""
    if tmin is None:
        trim_min = 0
    else:
        assert tmin >= dset_tmin, 'clip.tmin should be above dset.tmin'
        trim_min = int((-dset_tmin + tmin) * sample_rate)
    if tmax is None:
        trim_max = estimates.shape[-1]
    else:
        trim_max = int((-dset_tmin + tmax) * sample_rate)
    return estimates[..., trim_min:trim_max], candidates[..., trim_min:trim_max]


def clip_scores(estimates: torch.Tensor, candidates: torch.Tensor, pool: bool = False,
                center: bool = False) -> torch.Tensor:
    """bm/losses.py:77-95 ClipLoss.get_scores (after trimming; `linear` is dead code, :35/:82)."""
    if pool:
        estimates = estimates.mean(dim=2, keepdim=True)
        candidates = candidates.mean(dim=2, keepdim=True)
    if center:
        estimates = estimates - estimates.mean(dim=(1, 2), keepdim=True)
        candidates = candidates - candidates.mean(dim=(1, 2), keepdim=True)
    inv_norms = 1 / (1e-8 + candidates.norm(dim=(1, 2), p=2))
    return torch.einsum("bct,oct,o->bo", estimates, candidates, inv_norms)


def clip_probabilities(estimates, candidates, **kw) -> torch.Tensor:
    """bm/losses.py:97-102."""
    return F.softmax(clip_scores(estimates, candidates, **kw), dim=1)


def clip_loss(estimate: torch.Tensor, candidate: torch.Tensor, **kw) -> torch.Tensor:
    """bm/losses.py:104-114 ClipLoss.forward (mask must be all-ones, :110)."""
    assert estimate.size(0) <= candidate.size(0), "need at least as many targets as estimates"
    scores = clip_scores(estimate, candidate, **kw)
    target = torch.arange(len(scores))
    return F.cross_entropy(scores, target)


def clip_loss_symmetric(estimate: torch.Tensor, candidate: torch.Tensor, target_offset: int = 0, **kw) -> torch.Tensor:
    """NOT in the reference (its ClipLoss is the row term, bm/losses.py:104-114): the symmetric CLIP objective the
    hot-path contract names ("row/col softmax"), restated here as the checker of the opt-in ``symmetric`` extension.
    Row term as ``clip_loss``; column term: every target candidate classifies the B estimates."""
    scores = clip_scores(estimate, candidate, **kw)
    n = len(scores)
    target = torch.arange(n)
    rows = F.cross_entropy(scores, target + target_offset)
    cols = F.cross_entropy(scores[:, target_offset:target_offset + n].t(), target)
    return 0.5 * (rows + cols)


def clip_loss_symmetric_node(estimate_all: torch.Tensor, candidate_all: torch.Tensor, rank: int, B: int, **kw) -> torch.Tensor:
    """NOT in the reference: the symmetric objective of ``clip_loss_symmetric`` with whole-node negatives on both sides,
    as rank ``rank`` of a data-parallel run sees it -- its B estimates (rows [rank B, (rank + 1) B) of the gathered
    estimates) against every candidate, and its B target candidates against every estimate.  Checker of
    ``ClipLoss(symmetric=True)`` under ``negatives="node"`` (brainmagick_amd/losses.py)."""
    scores = clip_scores(estimate_all, candidate_all, **kw)             # [N B, N B], candidate norms applied
    own = torch.arange(B) + rank * B
    rows = F.cross_entropy(scores[own], own)
    cols = F.cross_entropy(scores[:, own].t(), own)
    return 0.5 * (rows + cols)


def topk_accuracy(probs: torch.Tensor, labels: torch.Tensor, row_labels: torch.Tensor,
                  topk: int = 10) -> float:
    """scripts/run_eval_probs.py:237-264 _get_accuracy_from_probs, segment-level: a row is a hit
    if its own label is among the labels of its ``topk`` most probable candidates."""
    idx = probs.topk(topk, dim=1, sorted=False).indices
    hit = (labels[idx] == row_labels[:, None]).any(1)
    return hit.float().mean().item()


def get_wer_loop(estimates, outputs, word_hashes, kept, topx=10):
    """bm/wer.py:71-120, the per-segment loop exactly as written (``kept`` = the indices drawn by the
    randperm at :72-73).  Returns {'wer', 'wer_vocab'}."""
    negatives = outputs[kept].clone()
    negative_hashes = word_hashes[kept].clone()
    correct = 0.
    correct_vocab = 0.
    for estimate, word_hash, output in zip(estimates, word_hashes, outputs):
        negatives[-1] = output
        negative_hashes[-1] = word_hash
        probas = clip_probabilities(estimate[None], negatives)[0]
        negative_hashes_vocab, indices = torch.unique(negative_hashes, return_inverse=True)
        probas_vocab = torch.zeros(len(negative_hashes_vocab), dtype=probas.dtype)
        probas_vocab.scatter_add_(0, indices, probas)
        _, bests = probas.topk(min(topx, len(probas)))
        _, bests_vocab = probas_vocab.topk(min(topx, len(probas_vocab)))
        correct += (negative_hashes[bests] == word_hash).any().item()
        correct_vocab += (negative_hashes_vocab[bests_vocab] == word_hash).any().item()
    correct /= len(estimates)
    correct_vocab /= len(estimates)
    return {'wer': 1 - correct, 'wer_vocab': 1 - correct_vocab}


# ----------------------------------------------------------------------------------------------
# ScaleReject front end
# ----------------------------------------------------------------------------------------------
def scale_reject(meg, features, recording_index, meg_center, meg_scale, feature_center=None,
                 feature_scale=None, limit=16, clip=False):
    """bm/norm.py:239-275 (BatchScaler._transform: per-segment RobustScaler.transform picked by
    recording, :86-87) followed by bm/norm.py:325-341 (ScaleReject.__call__).  Returns
    (meg, features, keep)."""
    out = []
    for entry_meg, entry_rec in zip(meg, recording_index):
        r = int(entry_rec)
        out.append(((entry_meg.t() - meg_center[r].to(entry_meg)) / meg_scale[r].to(entry_meg)).t())
    meg = torch.stack(out)
    if feature_center is not None:
        x = features.permute(0, 2, 1).reshape(-1, features.shape[1])          # _as_nd, norm.py:23-26
        x = (x - feature_center.to(x)) / feature_scale.to(x)
        features = x.view(features.shape[0], features.shape[2], -1).permute(0, 2, 1).contiguous()
    if clip:
        meg = meg.clamp(-limit, limit)
    reject = meg.abs().view(len(meg), -1).max(-1)[0] > limit
    keep = ~reject
    return meg[keep], features[keep], keep


# ----------------------------------------------------------------------------------------------
# Optimiser step and the solver's step body
# ----------------------------------------------------------------------------------------------
def adam_step(param, grad, exp_avg, exp_avg_sq, step: int, lr=3e-4, beta1=0.9, beta2=0.999,
              eps=1e-8):
    """torch.optim.Adam single-tensor update as configured at bm/train.py:118-119 (no weight
    decay, no amsgrad).  In-place on param / exp_avg / exp_avg_sq; ``step`` is 1-based."""
    exp_avg.lerp_(grad, 1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bias_correction1 = 1 - beta1 ** step
    bias_correction2 = 1 - beta2 ** step
    step_size = lr / bias_correction1
    denom = (exp_avg_sq.sqrt() / math.sqrt(bias_correction2)).add_(eps)
    param.addcdiv_(exp_avg, denom, value=-step_size)


def sync_gradients_reference(per_rank_grads: tp.List[tp.List[torch.Tensor]]):
    """flashy.distrib.sync_model as used at bm/solver.py:386 [upstream, unverified; parity
    unpinned]: every rank ends with the mean over ranks of each gradient."""
    world = len(per_rank_grads)
    return [sum(gs) / world for gs in zip(*per_rank_grads)]


class OracleModel:
    """Holds a reference-layout state_dict and runs the solver's step body
    (bm/solver.py:373-390: loss -> zero_grad -> backward -> step) on CPU with torch autograd."""

    def __init__(self, state_dict: dict, cfg: dict, hidden: int, out_channels: int,
                 dtype=torch.float32, lr=3e-4, betas=(0.9, 0.999), eps=1e-8):
        self.cfg = full_cfg(cfg)
        self.hidden = hidden
        self.out_channels = out_channels
        self.dtype = dtype
        self.sd = {}
        for k, v in state_dict.items():
            v = v.detach().clone()
            if v.is_floating_point():
                v = v.to(dtype)
            self.sd[k] = v
        self.param_names = [k for k in self.sd
                            if self.sd[k].is_floating_point() and "running_" not in k]
        self.lr, self.betas, self.eps = lr, betas, eps
        self.adam_state = {k: (torch.zeros_like(self.sd[k]), torch.zeros_like(self.sd[k]))
                           for k in self.param_names}
        self.step_count = 0

    def forward(self, meg, positions, subjects, training=False, ban_center=None,
                new_buffers=None):
        return simpleconv_forward(self.sd, self.cfg, meg.to(self.dtype), positions.to(self.dtype),
                                  subjects, self.hidden, self.out_channels, training,
                                  ban_center, new_buffers)

    def loss_and_grads(self, meg, positions, subjects, candidates, training=True,
                       ban_center=None, update_buffers=True):
        for k in self.param_names:
            self.sd[k].requires_grad_(True)
            self.sd[k].grad = None
        new_buffers: dict = {}
        est = self.forward(meg, positions, subjects, training, ban_center, new_buffers)
        loss = clip_loss(est, candidates.to(self.dtype))
        loss.backward()
        grads = {k: self.sd[k].grad.detach().clone() for k in self.param_names
                 if self.sd[k].grad is not None}
        for k in self.param_names:
            self.sd[k].requires_grad_(False)
            self.sd[k].grad = None
        if update_buffers:
            for k, v in new_buffers.items():
                self.sd[k] = v.detach()
        return loss.detach(), est.detach(), grads

    def apply_adam(self, grads):
        self.step_count += 1
        for k, g in grads.items():
            m, v = self.adam_state[k]
            adam_step(self.sd[k], g, m, v, self.step_count, self.lr, self.betas[0],
                      self.betas[1], self.eps)

    def train_step(self, meg, positions, subjects, candidates, ban_center=None):
        loss, est, grads = self.loss_and_grads(meg, positions, subjects, candidates, True,
                                               ban_center)
        self.apply_adam(grads)
        return loss, est, grads
