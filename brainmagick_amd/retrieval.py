"""Batched segment-retrieval evaluation on the HIP path -- the second half of the headline metric
("top-10 segment-retrieval accuracy").

Mirrors ``scripts/run_eval_probs.py``: ``builds_probs`` (:267-307, probability of every candidate
for every prediction, in batches) and ``_get_accuracy_from_probs`` (:237-264, a row is a hit when
its own label is among the labels of its top-k candidates).  Differences to the reference: the
candidate norms are computed once (not once per batch of predictions), probabilities stay on the
GPU, and top-k + label matching are one HIP kernel (``bm_topk_rows``) instead of
``topk`` + gather + compare.
"""
import typing as tp

import torch

from . import hip_ops as H
from .losses import ClipLoss


def builds_probs(clip: ClipLoss, preds: torch.Tensor, trues: torch.Tensor, dset_args=None,
                 batch_size: int = 1000, tmin=None, tmax=None) -> torch.Tensor:
    """[N, C, T] predictions x [N', C, T] candidates -> [N, N'] probabilities (on the GPU)."""
    trim_min = trim_max = None
    if tmin is not None:
        trim_min = int((tmin - dset_args.tmin) * dset_args.sample_rate)
    if tmax is not None:
        trim_max = int((tmax - dset_args.tmin) * dset_args.sample_rate)
    preds = preds[..., trim_min:trim_max]
    trues = trues[..., trim_min:trim_max]
    candidates = trues.cuda().contiguous()
    Bc = candidates.shape[0]
    K = candidates.numel() // Bc
    inv = H.clip_inv_norms(candidates)
    probs = torch.empty(len(preds), Bc, device=candidates.device, dtype=torch.float32)
    for lo in range(0, len(preds), batch_size):
        est = preds[lo:lo + batch_size].cuda().contiguous()
        part = H.gemm_nt_partials(est, candidates, 1, est.shape[0], Bc, K, (0, K), (0, K))
        _, p, _, _ = H.clip_ce(part, inv, want_probs=True)
        probs[lo:lo + batch_size] = p
    return probs


def get_accuracy_from_probs(probs: torch.Tensor, target_labels: torch.Tensor,
                            vocab_labels: torch.Tensor, topk: int = 10) -> float:
    """scripts/run_eval_probs.py:237-264.  probs [B, V]; target_labels [B]; vocab_labels [V]."""
    assert len(target_labels) == len(probs)
    assert len(vocab_labels) == probs.shape[1]
    _, _, hits = H.topk_rows(probs.contiguous(), topk,
                             vocab_labels.to(probs.device, torch.int64).contiguous(),
                             target_labels.to(probs.device, torch.int64).contiguous())
    return hits.float().mean().item()


def segment_topk_accuracy(clip: ClipLoss, preds: torch.Tensor, trues: torch.Tensor,
                          labels: tp.Optional[torch.Tensor] = None, topks=(1, 5, 10),
                          batch_size: int = 1000) -> tp.Dict[str, float]:
    """Top-k segment accuracy: segment i is retrieved when candidate i (or any candidate carrying
    the same label, e.g. the same audio segment hash) is among its k most probable candidates."""
    if labels is None:
        labels = torch.arange(len(trues))
    probs = builds_probs(clip, preds, trues, batch_size=batch_size)
    return {f"top{k}": get_accuracy_from_probs(probs, labels[:len(preds)], labels, k) for k in topks}
