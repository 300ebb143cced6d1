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


def get_wer(clip: ClipLoss, estimates: torch.Tensor, outputs: torch.Tensor, word_hashes: torch.Tensor,
            n_negatives: tp.Optional[int] = 10_000, topx: int = 10,
            generator: tp.Optional[torch.Generator] = None,
            batch_size: int = 1000) -> tp.Dict[str, float]:
    """Word-level top-k "word error rate" of bm/wer.py:67-121, batched on the GPU.

    Reference semantics, per test segment i: the negatives are a fixed random subset of the test
    outputs whose LAST entry is replaced by the segment's own target (wer.py:71-78,93-94); the
    probabilities over negatives are aggregated per word (wer.py:100-103); the segment is correct
    when its word is among the ``topx`` most probable negatives (``wer``) / vocabulary words
    (``wer_vocab``).  The reference re-scans all N candidates once per test segment (a Python loop
    of GEMVs); here: ONE MFMA GEMM for all scores, a row-wise dot product for the own-target column,
    a row softmax, a deterministic per-word segmented sum and two top-k passes."""
    n = len(outputs)
    if n_negatives:
        perm = torch.randperm(n, generator=generator)
        kept = perm[:n_negatives]
    else:
        kept = torch.arange(n)
    negatives = outputs[kept].cuda().contiguous()
    negative_hashes = word_hashes[kept].to(torch.int64)
    N = len(negatives)
    K = negatives.numel() // N
    dev = negatives.device
    # fixed columns 0..N-2 grouped by word; the own-target column N-1 is added per row
    fixed_hashes = negative_hashes[:N - 1]
    vocab, inverse = torch.unique(torch.cat([fixed_hashes, word_hashes.to(torch.int64)]),
                                  return_inverse=True)
    col_vocab = inverse[:N - 1]
    order = torch.argsort(col_vocab, stable=True).to(torch.int32)
    counts = torch.bincount(col_vocab, minlength=len(vocab))
    seg = torch.zeros(len(vocab) + 1, dtype=torch.int32)
    seg[1:] = torch.cumsum(counts, 0)
    own_vocab = inverse[N - 1:]                               # vocabulary index of every segment's word
    order_d, seg_d, vocab_d = order.to(dev), seg.to(dev), vocab.to(dev)
    inv = H.clip_inv_norms(negatives)
    correct = correct_vocab = 0.0
    for lo in range(0, len(estimates), batch_size):
        est = estimates[lo:lo + batch_size].cuda().contiguous()
        own = outputs[lo:lo + batch_size].cuda().contiguous()
        wh = word_hashes[lo:lo + batch_size].to(dev, torch.int64)
        m = est.shape[0]
        part = H.gemm_nt_partials(est, negatives, 1, m, N, K, (0, K), (0, K))
        scores, _, _, _ = H.clip_ce(part, inv)
        scores[:, N - 1] = H.rowwise_dot(est.view(m, K), own.view(m, K), H.clip_inv_norms(own))
        probas = H.row_softmax(scores)
        # candidate-level: labels of the columns, with the own word on the last column
        # (a per-row label for the last column: handle it by checking the two cases separately)
        sentinel = int(min(int(negative_hashes.min()), int(word_hashes.min()))) - 1
        col_labels = torch.cat([fixed_hashes.to(dev),
                                torch.full((1,), sentinel, dtype=torch.int64, device=dev)])
        idx, _, hits = H.topk_rows(probas, min(topx, N), col_labels, wh)
        own_in_top = (idx == N - 1).any(1)
        correct += float((hits.bool() | own_in_top).sum())
        # vocabulary-level
        pv = H.segment_sum_cols(probas, order_d, seg_d)
        ov = own_vocab[lo:lo + m].to(dev)
        pv[torch.arange(m, device=dev), ov] += probas[:, N - 1]
        _, _, hits_v = H.topk_rows(pv, min(topx, pv.shape[1]), vocab_d, wh)
        # vocabulary entries that have zero mass only exist for other rows' own words: harmless
        correct_vocab += float(hits_v.sum())
    total = len(estimates)
    return {"wer": 1 - correct / total, "wer_vocab": 1 - correct_vocab / total}
