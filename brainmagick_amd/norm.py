"""Host-side mirror of the per-batch part of ``bm/norm.py``: ``BatchScaler.transform`` +
``ScaleReject`` -- the step immediately in front of the model in ``Solver._process_batch``
(bm/solver.py:245-246), fused into one HIP streaming kernel per tensor.

Fitting the scalers (quantiles over 200 segments per recording, bm/norm.py:56-84,161-237) is a
one-off host job that stays with the reference; ``DeviceBatchScaler.from_reference`` converts a
fitted reference ``BatchScaler`` (its ``center_`` / ``scale_`` tensors) into device tables.
"""
import dataclasses
import typing as tp

import torch

from . import hip_ops as H


class DeviceBatchScaler:
    """Per-recording robust-scaler tables [R, C] for the MEG and (centre, scale) vectors [F] for
    the features, resident on the GPU."""

    def __init__(self, meg_center: torch.Tensor, meg_scale: torch.Tensor,
                 feature_center: tp.Optional[torch.Tensor] = None,
                 feature_scale: tp.Optional[torch.Tensor] = None, device="cuda"):
        self.meg_center = meg_center.to(device, torch.float32).contiguous()
        self.meg_scale = meg_scale.to(device, torch.float32).contiguous()
        self.feature_center = None if feature_center is None else \
            feature_center.to(device, torch.float32).contiguous()
        self.feature_scale = None if feature_scale is None else \
            feature_scale.to(device, torch.float32).contiguous()

    @classmethod
    def from_reference(cls, batch_scaler, n_channels: int, device="cuda"):
        """``batch_scaler``: a fitted reference ``bm.norm.BatchScaler`` (bm/norm.py:145-237)."""
        n_rec = max(batch_scaler.meg_scalers) + 1
        center = torch.zeros(n_rec, n_channels)
        scale = torch.ones(n_rec, n_channels)
        for idx, sc in batch_scaler.meg_scalers.items():
            center[idx, :len(sc.center_)] = sc.center_
            scale[idx, :len(sc.scale_)] = sc.scale_
        fdim = batch_scaler.features_builder.dimension
        fcenter, fscale = torch.zeros(fdim), torch.ones(fdim)
        for name, fs in batch_scaler.feature_scalers.items():
            if hasattr(fs, "center_"):
                sl = batch_scaler.features_builder.get_slice(name)
                fcenter[sl] = fs.center_
                fscale[sl] = fs.scale_
        return cls(center, scale, fcenter, fscale, device)

    def transform(self, batch):
        """BatchScaler.transform (bm/norm.py:239-275) on a device batch; returns a new batch."""
        meg, _ = H.center_scale(batch.meg.contiguous(), self.meg_center, self.meg_scale,
                                group=batch.recording_index.contiguous())
        features = batch.features
        if self.feature_center is not None:
            features, _ = H.center_scale(features.contiguous(), self.feature_center[None],
                                         self.feature_scale[None])
        return dataclasses.replace(batch, meg=meg, features=features)


def _subset(batch, keep: torch.Tensor):
    """``batch[keep]`` with the reference's semantics (bm/dataset.py:242-257): tensor fields are
    indexed, list fields (``_recordings``, ``_event_lists``) keep the selected items, empty lists stay
    empty.  Uses the batch's own ``__getitem__`` when it has one (the reference's SegmentBatch)."""
    if hasattr(type(batch), "__getitem__"):
        return batch[keep]
    idx = keep.nonzero().flatten()
    picked = idx.tolist()
    kw = {}
    for field in dataclasses.fields(batch):
        data = getattr(batch, field.name)
        if isinstance(data, list):
            kw[field.name] = [data[i] for i in picked] if data else []
        elif isinstance(data, torch.Tensor):
            kw[field.name] = data[idx]
        else:
            kw[field.name] = data
    return dataclasses.replace(batch, **kw)


class ScaleReject:
    """bm/norm.py:311-345.  Rescales MEG and features; rejects items whose scaled MEG still exceeds
    ``limit`` (or, with ``clip``, clamps instead).  With ``clip=True`` (conf/config.yaml:131) no
    amplitude rejection can occur, so the step runs without any host synchronisation; otherwise
    one small read-back of the per-segment maxima decides the (rare) compaction."""

    def __init__(self, scaler: DeviceBatchScaler, limit=16, exclude_empty_features=False, clip=False):
        self.scaler = scaler
        self.limit = limit
        self.clip = clip
        self.exclude_empty_features = exclude_empty_features
        self._rejection_count = 0
        self._count = 0

    def __call__(self, batch) -> tp.Tuple[tp.Any, torch.Tensor]:
        sc = self.scaler
        meg, maxabs = H.center_scale(batch.meg.contiguous(), sc.meg_center, sc.meg_scale,
                                     group=batch.recording_index.contiguous(), clip=self.clip,
                                     limit=float(self.limit), want_maxabs=not self.clip)
        features = batch.features
        if sc.feature_center is not None:
            features, _ = H.center_scale(features.contiguous(), sc.feature_center[None],
                                         sc.feature_scale[None])
        self._count += len(meg)
        keep = torch.ones(len(meg), dtype=torch.bool, device=meg.device)
        if maxabs is not None:
            keep &= ~(maxabs > self.limit)
        if self.exclude_empty_features:
            keep &= batch.features_mask.view(len(meg), -1).sum(-1) != 0
        batch = dataclasses.replace(batch, meg=meg, features=features)
        if maxabs is None and not self.exclude_empty_features:
            return batch, keep                       # nothing can be rejected: no sync
        n_reject = int((~keep).sum().item())
        self._rejection_count += n_reject
        if n_reject == 0:
            return batch, keep
        return _subset(batch, keep), keep

    @property
    def rejection_rate(self):
        return self._rejection_count / max(self._count, 1)
