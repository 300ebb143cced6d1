"""Host-side mirror of the hot loop of ``bm/solver.py``: ``_process_batch`` (bm/solver.py:230-321)
and the per-batch body of ``_run_one_epoch`` (bm/solver.py:343-390), for the decode task with the
CLIP loss -- the path BASELINE.json names.  Everything around it in the reference (flashy stages,
checkpoint commit, tensorboard, dataset construction) stays the reference's business.

Differences to the reference, all MI355X-motivated and opt-in/neutral:
  * gradients / optimizer: ONE flat bucket -> reduce-scatter + fused Adam on the shard + all-gather
    (``distrib.sharded_step``) instead of ``flashy.distrib.sync_model`` + ``optimizer.step()``;
  * ``negatives="node"``: candidates of all ranks are all-gathered over xGMI while the encoder runs
    (``negatives="local"`` reproduces the reference: negatives only within a GPU, README.md:139-143);
  * the loss stays on the device (no per-step ``.item()`` sync); read it at print points.
"""
import typing as tp

import torch

from . import distrib
from . import hip_ops as H
from .losses import ClipLoss
from .optim import FlatAdam


class Solver:
    def __init__(self, model: torch.nn.Module, loss: tp.Optional[ClipLoss] = None,
                 optimizer: tp.Optional[FlatAdam] = None, device: str = "cuda",
                 offset_meg_ms: float = 0., sample_rate: float = 120., negatives: str = "local",
                 lr: float = 3e-4, betas=(0.9, 0.999), scale_reject=None,
                 feature_model: tp.Optional[torch.nn.Module] = None, check_finite: bool = True):
        assert negatives in ("local", "node")
        self.device = torch.device(device)
        self.model = model.to(self.device)
        self.loss = (loss or ClipLoss()).to(self.device)
        self.feature_model = feature_model.to(self.device) if feature_model is not None else None
        world = distrib.world_size()
        if optimizer is None:
            params = list(self.model.parameters()) + list(self.loss.parameters())
            if self.feature_model is not None:
                params += list(self.feature_model.parameters())       # bm/train.py:116-117
            optimizer = FlatAdam(params, lr=lr, betas=betas, pad_to=max(world, 1) * 4)
        self.optimizer = optimizer
        self.offset_meg_ms = offset_meg_ms
        self.sample_rate = sample_rate
        self.negatives = negatives
        self._gather = distrib.CandidateGather() if negatives == "node" else None
        self.scale_reject = scale_reject          # brainmagick_amd.norm.ScaleReject or None
        if negatives == "node" and scale_reject is not None and \
                (not scale_reject.clip or scale_reject.exclude_empty_features):
            # per-rank rejection would leave ranks with different numbers of segments: the candidate
            # all-gather (equal blocks, target_offset = rank * B) cannot express that
            raise ValueError('negatives="node" needs the same number of segments on every rank: use '
                             'ScaleReject(clip=True, exclude_empty_features=False) (conf/config.yaml:131 '
                             'norm.clip=true) or negatives="local"')
        # flashy.distrib.sync_model also averages the float buffers (BatchNorm running statistics)
        self._buffers = distrib.BufferBucket(self._all_models())
        self.check_finite = check_finite          # bm/solver.py:258-260 asserts (one fused host sync)
        self.loss.defer_mask_check = check_finite  # bm/losses.py:110 assert: same sync point, one step late
        self._last_batch = None
        self._prefetched = None

    # -- bm/solver.py:230-321 ------------------------------------------------------------------
    def _prepare(self, batch):
        """Everything of ``_process_batch`` in front of the model: device copy, scale / clamp / reject, the
        finiteness asserts, offset slicing -- and, with whole-node negatives and constant candidates, the start of
        the candidate all-gather (candidates are inputs: the exchange runs next to the encoder)."""
        batch = batch.to(self.device)
        if self.scale_reject:
            batch, reject_mask = self.scale_reject(batch)          # bm/solver.py:245-246
        else:
            reject_mask = torch.ones(len(batch.meg), dtype=torch.bool, device=self.device)
        meg = batch.meg
        features = batch.features
        features_mask = batch.features_mask
        if len(meg) == 0:
            return None
        if self.check_finite:
            # bm/solver.py:258-260 (three separate asserts / syncs in the reference, one here, in _check_flags).
            # The finiteness test rides on the max|x| pass that the f16x2 contractions need anyway (the maxima stay
            # attached to the tensors); in the other compute modes it is the one pass over each tensor.
            flag = H.index_error_flag(meg.device)
            if meg.is_contiguous() and features.is_contiguous() and meg.dtype == features.dtype == torch.float32:
                H.amax(meg, nonfinite_flag=flag[1:2])
                # candidates: norms, maximum and finiteness in one pass (ClipLoss finds them on the tensor)
                H.clip_inv_norms(features, nonfinite_flag=flag[1:2])
            else:
                finite = torch.isfinite(meg).all() & torch.isfinite(features).all()
                flag[1:2].bitwise_or_((~finite).to(torch.int32).view(1))
        if self.offset_meg_ms:
            # bm/solver.py:262-274: brain responses lag the audio by ~150 ms
            offset = int(self.offset_meg_ms / 1000 * self.sample_rate)
            meg = meg[..., offset:]
            features = features[..., :-offset]
            features_mask = features_mask[..., :-offset]
        features = features.contiguous()
        if self._gather is not None and self.feature_model is None:
            # candidates do not depend on the model: start the xGMI all-gather before the encoder
            self._gather.start(features)
        return batch, meg.contiguous(), features, features_mask, reject_mask

    def prefetch(self, next_batch) -> None:
        """Optional: hand over the batch of the NEXT step (``train_step(batch, next_batch=...)`` calls this between
        the loss and the backward pass).  Its preparation (device copy, scaling, the max|x| / finiteness pass) is
        enqueued now and -- the point -- with whole-node negatives its candidate all-gather (3 GB at 8 x 256
        wav2vec2-sized candidates, ~10 ms on xGMI against a ~5 ms forward) runs on the side stream NEXT TO THIS STEP'S
        BACKWARD instead of in front of the next forward.  Nothing here synchronises the host (the asserts of the
        prepared batch are read when it is consumed).  The next ``train_step`` / ``_process_batch`` recognises the
        batch by identity.  Every rank must prefetch (or not) alike."""
        if next_batch is None:
            self._prefetched = None
            return
        self._prefetched = (next_batch, self._prepare(next_batch))

    def _check_flags(self):
        """The ONE host synchronisation of a step: the device-side flag word holds "non-finite input"
        (bm/solver.py:258-260), "ClipLoss mask not all-true" (bm/losses.py:110) and "subject / layout index out of
        range" (the reference's `weights.gather` would have raised, bm/models/common.py:57) -- the last two raised by
        the PREVIOUS step."""
        if not self.check_finite:
            return
        flag = H.index_error_flag(self.device)
        index_err, nonfinite, bad_mask = flag.tolist()
        if nonfinite:
            flag[1:2].zero_()
            raise AssertionError("non-finite values in the MEG or feature tensors")
        if bad_mask:
            flag[2:3].zero_()
            raise AssertionError("mask is not supported for now (bm/losses.py:110; reported one step late)")
        if index_err:
            H.raise_if_index_error(self.device)

    def _process_batch(self, batch, training: bool = False):
        pre = self._prefetched
        self._prefetched = None
        if pre is not None and pre[0] is batch:
            prepared = pre[1]
        else:
            if pre is not None and self._gather is not None:
                self._gather.cancel()                     # a prefetched gather nobody will consume
            prepared = self._prepare(batch)
        if prepared is None:
            return None, None, None, None
        self._check_flags()
        batch, meg, features, features_mask, reject_mask = prepared
        inputs = dict(meg=meg)
        estimate = self.model(inputs, batch)
        if self.feature_model is not None:
            features = self.feature_model(features)                   # bm/solver.py:304-320
        return estimate, features, features_mask, reject_mask

    def _all_models(self):
        return [self.model] + ([self.feature_model] if self.feature_model is not None else [])

    def _candidates(self, output):
        """(candidates, target_offset): local, gathered on the side stream (constant candidates) or
        gathered with an autograd-aware all-gather (learnable candidates)."""
        if self.negatives != "node":
            return output, 0
        if self.feature_model is None:
            return self._gather.wait()
        return distrib.gather_learnable_candidates(output)

    # -- bm/solver.py:343-390 (one iteration) -----------------------------------------------------
    def train_step(self, batch, next_batch=None) -> torch.Tensor:
        """One iteration.  ``next_batch`` (optional): see ``prefetch``."""
        for m in self._all_models():
            m.train(True)
        self.loss.train(True)
        estimate, output, features_mask, _ = self._process_batch(batch, training=True)
        if estimate is None:
            # bm/solver.py:345-352: a fully rejected batch re-uses the last good one so that every
            # rank keeps issuing the same collectives
            if self._last_batch is None:
                raise RuntimeError("Empty batch and last batch is none")
            estimate, output, features_mask, _ = self._process_batch(self._last_batch, training=True)
        else:
            self._last_batch = batch
        output, target_offset = self._candidates(output)
        loss = self.loss(estimate, output, features_mask, target_offset=target_offset)
        if next_batch is not None:
            self.prefetch(next_batch)       # the next step's candidate all-gather runs next to this backward
        # bm/solver.py:375-380: `training_penalty` of ChannelMerger is identically 0 on this path
        # (merger_penalty > 0 is rejected at construction) and optim.svd defaults to 0.
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.collect_grads()      # one multi-tensor copy instead of an accumulate-add per parameter
        distrib.sharded_step(self.optimizer, self._buffers)
        return loss.detach()

    # -- checkpoint (bm/solver.py:64,115-117: flashy's commit writes the registered state on rank 0) --------
    def state_dict(self) -> dict:
        """COLLECTIVE in a data-parallel run (every rank calls it; rank 0 then saves): the Adam moments are sharded
        over the ranks and are gathered first -- ``FlatAdam.state_dict()`` itself never communicates."""
        self.optimizer.gather_moments()
        out = {"model": self.model.state_dict(), "optimizer": self.optimizer.state_dict()}
        if self.feature_model is not None:
            out["feature_model"] = self.feature_model.state_dict()
        return out

    def load_state_dict(self, state: dict) -> None:
        self.model.load_state_dict(state["model"])
        if self.feature_model is not None and "feature_model" in state:
            self.feature_model.load_state_dict(state["feature_model"])
        self.optimizer.load_state_dict(state["optimizer"])
        H.weights_changed()

    @torch.no_grad()
    def eval_step(self, batch) -> torch.Tensor:
        for m in self._all_models():
            m.train(False)
        self.loss.train(False)
        estimate, output, features_mask, _ = self._process_batch(batch, training=False)
        output, target_offset = self._candidates(output)
        return self.loss(estimate, output, features_mask, target_offset=target_offset)

    @torch.no_grad()
    def predict(self, batch):
        """(estimate, candidates) for retrieval evaluation (bm/wer.py:52, run_eval_probs.py:102)."""
        for m in self._all_models():
            m.train(False)
        estimate, output, _, _ = self._process_batch(batch, training=False)
        if self._gather is not None and self.feature_model is None:
            self._gather.wait()
        return estimate, output
