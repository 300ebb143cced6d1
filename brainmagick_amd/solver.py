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
                 feature_model: tp.Optional[torch.nn.Module] = None, check_finite: bool = True,
                 n_negatives: tp.Optional[int] = None, negative_pool_size: tp.Optional[int] = None,
                 batch_size: tp.Optional[int] = None):
        """``batch_size``: the configured per-rank batch size (`optim.batch_size` / world size, bm/train.py:37-39).
        Only whole-node negatives next to per-rank rejection need it: it is the block every rank contributes to the
        candidate all-gather.  Without it the block is the local batch's length and is verified over the ranks each
        step (one tiny all-reduce + read-back; that mode reads back its rejection count anyway)."""
        assert negatives in ("local", "node")
        self.batch_size = batch_size
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None and torch.cuda.is_available():
            self.device = torch.device("cuda", torch.cuda.current_device())      # comparable with tensor.device
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
        # Per-rank rejection (ScaleReject(clip=False) / exclude_empty_features) leaves ranks with different numbers
        # of segments.  Whole-node negatives then gather equal-sized blocks of the NOMINAL batch size, the rows a rank
        # rejected are padding, and ClipLoss masks them (candidate_valid) -- decided from the configuration, so every
        # rank takes the same branch.
        self._ragged_node = negatives == "node" and scale_reject is not None and \
            (not scale_reject.clip or scale_reject.exclude_empty_features)
        if self._ragged_node and feature_model is not None:
            raise ValueError('negatives="node" with a learnable feature model needs the same number of segments on '
                             'every rank: use ScaleReject(clip=True, exclude_empty_features=False) or negatives="local"')
        # bm/solver.py:153-165,358-371: `optim.negatives` -- candidates completed with random draws from a pool of
        # the previous batches' candidates (kept on the device here; the reference keeps it on the CPU)
        self.n_negatives = n_negatives
        if n_negatives is not None:
            if negatives == "node":
                raise ValueError('n_negatives (bm optim.negatives) completes the LOCAL candidates from a pool; '
                                 'it does not combine with negatives="node"')
            if negative_pool_size is None:
                negative_pool_size = 2 * n_negatives                    # bm/solver.py:155-157
            assert negative_pool_size >= n_negatives, "Pool of negatives should be larger than the number of negatives"
        self.negative_pool_size = negative_pool_size
        self.negative_pool: tp.Dict[str, tp.Optional[torch.Tensor]] = {"train": None, "valid": None}
        self.negative_generator: tp.Optional[torch.Generator] = None    # CPU generator of the pool draws (tests seed it)
        # flashy.distrib.sync_model also averages the float buffers (BatchNorm running statistics)
        self._buffers = distrib.BufferBucket(self._all_models())
        self.check_finite = check_finite          # bm/solver.py:258-260 asserts (one fused host sync)
        self.loss.defer_mask_check = check_finite  # bm/losses.py:110 assert: same sync point, one step late
        self._last_batch = None
        self._prefetched = None
        self._staged: tp.Dict[int, tuple] = {}    # id(host batch) -> (host batch, device batch): see stage()
        self._copy_stream = None
        self._substituted = False
        self._flag_ring: tp.List[torch.Tensor] = []   # pinned host copies of the device flag word (see _post_flags)
        self._flag_i = 0
        self._flag_ticket = None

    # -- bm/solver.py:243 `batch.to(self.device)` ----------------------------------------------------
    def stage(self, batch):
        """Host -> device copy of a batch on a dedicated copy stream, without blocking the host when the host
        tensors are pinned (``SegmentBatch.pin()``, or a DataLoader with ``pin_memory=True``): 121 MB per step at
        cfg2 travel next to the previous step's kernels instead of in front of this step's (the reference's
        ``batch.to(device)`` is a blocking pageable copy, bm/solver.py:243).  ``train_step(batch, next_batch=...)``
        stages ``next_batch`` before it enqueues anything of the current step; a batch that is already on the device
        passes through.  The compute stream waits for the copy when the batch is consumed (``_prepare``)."""
        if not isinstance(getattr(batch, "meg", None), torch.Tensor) or batch.meg.device == self.device:
            return batch
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._copy_stream):
            try:
                dev = batch.to(self.device, non_blocking=True)
            except TypeError:                       # the reference's SegmentBatch.to(device) has no such argument
                dev = batch.to(self.device)
        if len(self._staged) >= 4:                  # batches staged and never consumed
            self._staged.clear()
        self._staged[id(batch)] = (batch, dev)
        return dev

    def _on_device(self, batch):
        staged = self._staged.pop(id(batch), None)
        if staged is not None and staged[0] is batch:
            dev = staged[1]
        else:
            dev = self.stage(batch)
            if dev is batch:
                return batch
            self._staged.pop(id(batch), None)
        # the copies were enqueued on the copy stream: order them in front of the consumers, and keep the allocator
        # from recycling the blocks while the compute stream still reads them
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self._copy_stream)
        for name in ("meg", "features", "features_mask", "subject_index", "recording_index"):
            t = getattr(dev, name, None)
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(cur)
        return dev

    # -- bm/solver.py:230-321 ------------------------------------------------------------------
    def _prepare(self, batch):
        """Everything of ``_process_batch`` in front of the model: device copy, scale / clamp / reject, the
        finiteness asserts, offset slicing -- and, with whole-node negatives and constant candidates, the start of
        the candidate all-gather (candidates are inputs: the exchange runs next to the encoder)."""
        nominal = len(batch.meg)                   # segments before rejection: the block size of the candidate gather
        batch = self._on_device(batch)
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
            if self._ragged_node:
                if self.batch_size is not None and nominal > self.batch_size:
                    raise ValueError(f"batch of {nominal} segments exceeds Solver(batch_size={self.batch_size})")
                self._gather.start(features, block_rows=self.batch_size or nominal,
                                   verify_block=self.batch_size is None)
            else:
                self._gather.start(features)
        return batch, meg.contiguous(), features, features_mask, reject_mask

    def prefetch(self, next_batch) -> None:
        """Optional: hand over the batch of the NEXT step (``train_step(batch, next_batch=...)`` calls this between
        the loss and the backward pass).  Its preparation (scaling, the max|x| / finiteness pass; the device copy was
        staged at the start of the step) is enqueued now and -- the point -- with whole-node negatives its candidate
        all-gather (3 GB at 8 x 256 wav2vec2-sized candidates, ~10 ms on xGMI against a ~5 ms forward) runs on the
        side stream NEXT TO THIS STEP'S BACKWARD instead of in front of the next forward.  The host is not
        synchronised here unless a ``ScaleReject`` that can reject has to count its rejections (bm/norm.py:341, one
        small read-back); the asserts of the prepared batch are read when it is consumed.  The next ``train_step`` /
        ``_process_batch`` recognises the batch by identity.  Every rank must prefetch (or not) alike."""
        if next_batch is None:
            self._prefetched = None
            return
        prepared = self._prepare(next_batch)
        if prepared is None and self._gather is not None and self.feature_model is None and \
                self._last_batch is not None:
            # fully rejected: the step will re-run the last good batch (bm/solver.py:345-352).  Prepare THAT now, so
            # that this rank issues its candidate all-gather at the same place in the collective sequence as the others
            prepared = self._prepare(self._last_batch)
            self._prefetched = (next_batch, prepared, self._last_batch)
            return
        self._prefetched = (next_batch, prepared, next_batch)

    def _post_flags(self):
        """Enqueue an asynchronous read of the flag word (pinned host buffer + event) at this point of the stream:
        ``train_step`` evaluates it right before the backward pass, when the copy -- queued behind the previous step and
        this batch's preparation -- has long landed, instead of draining the whole queue with a blocking read in front of
        the forward pass (which cost every step the host's run-ahead: ~0.3 ms of idle GPU while the first launches of
        the step were being issued).  Nothing a non-finite batch could poison has happened by then: BatchNorm leaves
        its running estimates alone on non-finite statistics, and the optimizer has not stepped."""
        if not self.check_finite:
            return None
        if not self._flag_ring:
            self._flag_ring = [torch.empty(3, dtype=torch.int32).pin_memory() for _ in range(4)]
        buf = self._flag_ring[self._flag_i % len(self._flag_ring)]
        self._flag_i += 1
        buf.copy_(H.index_error_flag(self.device), non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        return buf, event

    def _check_flags(self, ticket=None):
        """The host's look at the device-side flag word: "non-finite input" (bm/solver.py:258-260), "ClipLoss mask not
        all-true" (bm/losses.py:110) and "subject / layout index out of range" (the reference's `weights.gather` would
        have raised, bm/models/common.py:57) -- the last two raised by the PREVIOUS step (``check_pending_flags()`` reads
        them without waiting for a next one).  With a ``ticket`` (``_post_flags``) only that earlier copy is waited
        for; without one this is a blocking read (evaluation, direct ``_process_batch`` callers)."""
        if not self.check_finite:
            return
        flag = H.index_error_flag(self.device)
        if ticket is not None:
            buf, event = ticket
            event.synchronize()
            index_err, nonfinite, bad_mask = buf.tolist()
        else:
            index_err, nonfinite, bad_mask = flag.tolist()
        if nonfinite:
            flag[1:2].zero_()
            raise AssertionError("non-finite values in the MEG or feature tensors")
        if bad_mask:
            flag[2:3].zero_()
            raise AssertionError("mask is not supported for now (bm/losses.py:110; reported one step late)")
        if index_err:
            H.raise_if_index_error(self.device)

    def check_pending_flags(self) -> None:
        """Raise what the LAST step left in the device-side flag word (mask not all-true, index out of range): call
        it after the final ``train_step`` / at the end of an evaluation loop -- the deferred asserts of a step are
        otherwise only read by the next one.  ``eval_step``, ``predict`` and ``state_dict`` call it."""
        self._check_flags()

    def _process_batch(self, batch, training: bool = False, defer_flags: bool = False):
        """bm/solver.py:230-321.  ``defer_flags`` (train_step): the asserts are posted here and evaluated before the
        backward pass (``_post_flags``); other callers get the reference's behaviour -- they raise before the model runs."""
        pre = self._prefetched
        self._prefetched = None
        if pre is not None and pre[0] is batch:
            prepared = pre[1]
            if pre[2] is not batch:                       # fully rejected at prefetch time: the last good batch
                batch = pre[2]                            # was prepared in its place (see prefetch)
                self._substituted = True
        else:
            if pre is not None and self._gather is not None:
                self._gather.cancel()                     # a prefetched gather nobody will consume
            prepared = self._prepare(batch)
        if prepared is None:
            return None, None, None, None
        if defer_flags:
            self._flag_ticket = self._post_flags()
        else:
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
        """(candidates, target_offset, candidate_valid | None): local, gathered on the side stream (constant
        candidates) or gathered with an autograd-aware all-gather (learnable candidates)."""
        if self.negatives != "node":
            return output, 0, None
        if self.feature_model is None:
            return self._gather.wait()
        return distrib.gather_learnable_candidates(output) + (None,)

    def _gathered_estimates(self, estimate):
        """``ClipLoss(symmetric=True)`` with whole-node negatives: the column term needs every rank's estimates (the
        all-gather of brain embeddings north_star names; its adjoint reduce-scatters their gradients back)."""
        if self.negatives == "node" and getattr(self.loss, "symmetric", False):
            return {"estimate_all": distrib.gather_learnable_candidates(estimate)[0]}
        return {}

    def _complete_with_pool(self, output, training: bool):
        """bm/solver.py:358-371: with ``optim.negatives`` set, fewer candidates than that are completed with a random
        draw from the pool of earlier candidates, and the pool takes the completed set in front."""
        if self.n_negatives is None:
            return output
        with torch.no_grad():
            if len(output) < self.n_negatives:
                phase = "train" if training else "valid"
                buf = self.negative_pool[phase]
                if buf is None:
                    buf = output.new_zeros((0,) + tuple(output.shape[1:]))
                n_kept = self.n_negatives - len(output)
                kept = torch.randperm(len(buf), generator=self.negative_generator)[:n_kept]
                output = torch.cat([output, buf[kept.to(buf.device)]], dim=0)
                buf = torch.cat([output.detach(), buf])
                self.negative_pool[phase] = buf[:self.negative_pool_size]
        return output

    # -- bm/solver.py:343-390 (one iteration) -----------------------------------------------------
    def train_step(self, batch, next_batch=None) -> torch.Tensor:
        """One iteration.  ``next_batch`` (optional): the batch of the next step -- its host -> device copy is
        staged right away on the copy stream (``stage``), the rest of its preparation and its candidate all-gather
        are enqueued between the loss and the backward pass (``prefetch``)."""
        if next_batch is not None:
            self.stage(next_batch)
        for m in self._all_models():
            m.train(True)
        self.loss.train(True)
        self._substituted = False
        # what the asserts of bm/solver.py:258-260 must leave untouched when they fire (the reference raises before it
        # touches any state; here they are evaluated after the forward pass, so that state is put back: `_rollback`)
        undo = self._snapshot()
        estimate, output, features_mask, _ = self._process_batch(batch, training=True, defer_flags=True)
        if estimate is None:
            # bm/solver.py:345-352: a fully rejected batch re-uses the last good one so that every
            # rank keeps issuing the same collectives
            if self._last_batch is None:
                raise RuntimeError("Empty batch and last batch is none")
            estimate, output, features_mask, _ = self._process_batch(self._last_batch, training=True,
                                                                     defer_flags=True)
        elif not self._substituted:
            self._last_batch = batch
        output, target_offset, valid = self._candidates(output)
        output = self._complete_with_pool(output, training=True)
        loss = self.loss(estimate, output, features_mask, target_offset=target_offset, candidate_valid=valid,
                         **self._gathered_estimates(estimate))
        if next_batch is not None:
            self.prefetch(next_batch)       # the next step's candidate all-gather runs next to this backward
        # bm/solver.py:375-380: `training_penalty` of every module that has one (ChannelMerger with merger_penalty:
        # a constant, see models/common.py); optim.svd defaults to 0 and stays with the reference
        for mod in self.model.modules():
            if hasattr(mod, "training_penalty"):
                loss = loss + mod.training_penalty.to(loss.device)
        # the asserts of bm/solver.py:258-260 (this batch) and of the previous step: before anything is updated
        ticket, self._flag_ticket = self._flag_ticket, None
        try:
            self._check_flags(ticket)
        except (AssertionError, IndexError):
            self._rollback(undo)
            raise
        self.optimizer.zero_grad(set_to_none=True)
        with self.optimizer.writing_grads():       # the weight-gradient kernels write straight into the flat bucket
            loss.backward()
        self.optimizer.collect_grads()      # one multi-tensor copy instead of an accumulate-add per parameter
        distrib.sharded_step(self.optimizer, self._buffers)
        return loss.detach()

    def _snapshot(self):
        """What a deferred assert must be able to put back (``_rollback``): references to the last good batch and the
        negatives pool, the state of the pool's generator, ONE small device copy of the flat BatchNorm buffer
        bucket (6 400 floats for the paper model) and one of the BatchNorm batch counters."""
        if not self.check_finite:
            return None
        flat = self._buffers.flat
        if flat is not None:
            if getattr(self, "_buffers_undo", None) is None or self._buffers_undo.shape != flat.shape:
                self._buffers_undo = torch.empty_like(flat)
            self._buffers_undo.copy_(flat)
        gen = self.negative_generator.get_state() if (self.n_negatives is not None and
                                                      self.negative_generator is not None) else None
        # BatchNorm batch counters: copied, not decremented on the way back -- a module that did not run in train mode
        # this step (unused branch, an eval-mode feature model) must keep its count
        counters = [mod.num_batches_tracked for model in self._all_models() for mod in model.modules()
                    if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm) and mod.num_batches_tracked is not None]
        counts = torch.stack(counters).clone() if counters else None
        return self._last_batch, dict(self.negative_pool), gen, counters, counts

    def _rollback(self, undo) -> None:
        """A deferred assert fired after the forward pass of ``train_step``: restore what the reference, which asserts
        before it touches anything (bm/solver.py:258-260), would not have touched -- the "last good batch" (a later
        fully rejected batch must not re-train on the poisoned one), the negatives pool (it would keep the non-finite
        candidates) and its generator, the BatchNorm running statistics and batch counters.  Gradients were not
        computed and the optimizer has not stepped.  (Independently of this, ``bn_finalize`` leaves the running
        statistics alone when the batch statistics are not finite -- a deliberate deviation from ``nn.BatchNorm1d``,
        which would poison its running estimates for good; README "Deviations".)"""
        if undo is None:
            return
        rejected = self._last_batch
        self._last_batch, self.negative_pool, gen, counters, counts = undo
        if gen is not None:
            self.negative_generator.set_state(gen)
        if self._buffers.flat is not None:
            self._buffers.flat.copy_(self._buffers_undo)
        if counts is not None:
            for counter, value in zip(counters, counts.unbind(0)):
                counter.copy_(value)
        # a prefetch made between the loss and this assert may have prepared the poisoned batch as the stand-in of a
        # fully rejected next batch (prefetch() substitutes `_last_batch`, which was the rejected one at that point):
        # a caller that catches the assert and goes on must not train on it
        pre = self._prefetched
        if pre is not None and pre[2] is not pre[0] and pre[2] is rejected and rejected is not self._last_batch:
            if self._gather is not None:
                self._gather.cancel()
            self._prefetched = None

    # -- checkpoint (bm/solver.py:64,115-117: flashy's commit writes the registered state on rank 0) --------
    def state_dict(self) -> dict:
        """COLLECTIVE in a data-parallel run (every rank calls it; rank 0 then saves): the Adam moments are sharded
        over the ranks and are gathered first -- ``FlatAdam.state_dict()`` itself never communicates."""
        self.check_pending_flags()
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
        output, target_offset, valid = self._candidates(output)
        output = self._complete_with_pool(output, training=False)
        loss = self.loss(estimate, output, features_mask, target_offset=target_offset, candidate_valid=valid,
                         **self._gathered_estimates(estimate))
        # the mask assert of THIS call is deferred to the flag word: an evaluation loop has no next train_step that
        # would read it (one small read-back per evaluation batch; the reference's assert synchronises as well)
        self.check_pending_flags()
        return loss

    @torch.no_grad()
    def predict(self, batch):
        """(estimate, candidates) for retrieval evaluation (bm/wer.py:52, run_eval_probs.py:102)."""
        for m in self._all_models():
            m.train(False)
        estimate, output, _, _ = self._process_batch(batch, training=False)
        if self._gather is not None and self.feature_model is None:
            self._gather.wait()
        self.check_pending_flags()
        return estimate, output
