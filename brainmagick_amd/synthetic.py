"""Seeded synthetic (sensors x time) segment batches shaped like BASELINE.json's configs.

There is no network / dataset here, so both the parity tests and ``bench.py`` feed the hot path
with synthetic batches that mirror what ``Solver._process_batch`` hands to the model
(bm/solver.py:243-286): ``meg`` already robust-scaled and clamped to +-20 (bm/norm.py:332-333),
``features`` standardised, an all-ones mask, int64 subject indices and per-recording 2-D sensor
layouts in [0,1]^2 with INVALID=-0.1 for padded sensors (bm/models/common.py:188,214,228).

All randomness comes from a CPU ``torch.Generator`` so the very same tensors can be fed to the
CPU oracle and to the HIP path.
"""
import dataclasses
import typing as tp

import torch

INVALID = -0.1

# name -> (C sensors, T samples, F feature dim, n_subjects, batch)   [SURVEY.md §8d]
CONFIGS: tp.Dict[str, tp.Dict[str, int]] = {
    "cfg1": dict(C=273, T=360, F=120, S=4, B=16),       # bm/mockdata fake study, CPU reference case
    "cfg2": dict(C=208, T=360, F=120, S=27, B=256),     # gwilliams2022-shaped, mel features
    "cfg3": dict(C=208, T=360, F=1024, S=27, B=256),    # same, wav2vec2-style features
    "cfg5": dict(C=273, T=360, F=1024, S=115, B=256),   # audio_mous MEG + broderick2019 EEG mixed
}


@dataclasses.dataclass
class Recording:
    """The fields of bm.studies.api.Recording that the model touches (common.py:190-233)."""
    recording_index: int
    layout: torch.Tensor          # [n_channels, 2] positions in [0,1]^2
    study: str = "synthetic"

    @property
    def recording_uid(self) -> str:
        return f"{self.study}_{self.recording_index}"

    def study_name(self) -> str:
        return self.study


@dataclasses.dataclass
class SegmentBatch:
    """Field-compatible subset of bm.dataset.SegmentBatch (bm/dataset.py:209-278)."""
    meg: torch.Tensor             # [B, C, T] fp32
    features: torch.Tensor        # [B, F, T] fp32
    features_mask: torch.Tensor   # [B, 1, T] bool
    subject_index: torch.Tensor   # [B] int64
    recording_index: torch.Tensor  # [B] int64
    _recordings: tp.List[Recording] = dataclasses.field(default_factory=list)
    _event_lists: tp.List[tp.Any] = dataclasses.field(default_factory=list)

    def __len__(self) -> int:
        return len(self.meg)

    def to(self, device, non_blocking: bool = False) -> "SegmentBatch":
        """bm/dataset.py:259-266.  ``non_blocking`` (extension): asynchronous copies when the host tensors are pinned
        (``pin()``), enqueued on the current stream -- ``Solver.stage`` runs them on its copy stream."""
        kw = {}
        for field in dataclasses.fields(self):
            data = getattr(self, field.name)
            kw[field.name] = data.to(device, non_blocking=non_blocking) if isinstance(data, torch.Tensor) else data
        return SegmentBatch(**kw)

    def pin(self) -> "SegmentBatch":
        """The same batch with its host tensors in page-locked memory (what a DataLoader with ``pin_memory=True``
        delivers): the precondition of an asynchronous host -> device copy."""
        kw = {}
        for field in dataclasses.fields(self):
            data = getattr(self, field.name)
            kw[field.name] = data.pin_memory() if isinstance(data, torch.Tensor) and not data.is_cuda else data
        return SegmentBatch(**kw)

    def replace(self, **kwargs) -> "SegmentBatch":
        return dataclasses.replace(self, **kwargs)

    def __getitem__(self, index) -> "SegmentBatch":
        """bm/dataset.py:242-257: tensors are indexed, list fields keep the selected items."""
        picked = torch.arange(len(self), device=self.meg.device)[index].tolist()
        kw = {}
        for field in dataclasses.fields(self):
            data = getattr(self, field.name)
            if isinstance(data, list):
                kw[field.name] = [data[i] for i in picked] if data else []
            else:
                kw[field.name] = data[index]
        return SegmentBatch(**kw)

    def positions(self) -> torch.Tensor:
        """What PositionGetter.get_positions returns (common.py:225-233): [B, C, 2], CPU."""
        B, C, _ = self.meg.shape
        pos = torch.full((B, C, 2), INVALID)
        for i, rec in enumerate(self._recordings):
            pos[i, :len(rec.layout)] = rec.layout
        return pos


def make_layouts(n_layouts: int, n_channels: tp.Sequence[int], gen: torch.Generator,
                 study: str = "synthetic") -> tp.List[Recording]:
    return [Recording(i, torch.rand(n_channels[i % len(n_channels)], 2, generator=gen), study)
            for i in range(n_layouts)]


def make_batch(B: int, C: int, T: int, F: int, S: int, seed: int = 2036, n_layouts: int = 1,
               mixed_eeg: bool = False, planted: bool = False,
               recordings: tp.Optional[tp.List[Recording]] = None,
               world_seed: int = 1234, noise: float = 0.5) -> SegmentBatch:
    """Seed 2036 = conf/config.yaml:33.  ``mixed_eeg`` reproduces cfg5: each sample is either a
    273-sensor MEG recording or a 128-sensor EEG recording zero-padded to C (dataset.py:353-354),
    EEG subjects 0..18 and MEG subjects 19..S-1.  ``planted`` adds a shared latent so that the
    contrastive task is learnable (used for the loss-curve / top-10 parity runs); the mixing
    matrices and the sensor layouts of the planted "world" only depend on ``world_seed`` so that
    train and held-out batches (different ``seed``) share them; ``noise`` is the standard deviation of
    the additive sensor / feature noise (0.5 = easy task, larger values push the retrieval accuracy
    away from saturation)."""
    gen = torch.Generator().manual_seed(seed)
    if recordings is None and planted:
        wgen = torch.Generator().manual_seed(world_seed + 1)
        recordings = make_layouts(n_layouts, [C], wgen)
    if recordings is None:
        if mixed_eeg:
            recordings = make_layouts(max(n_layouts, 2), [C, 128], gen)
        else:
            recordings = make_layouts(n_layouts, [C], gen)
    rec_idx = torch.randint(0, len(recordings), (B,), generator=gen)
    if mixed_eeg:
        is_eeg = torch.tensor([len(recordings[int(i)].layout) < C for i in rec_idx])
        n_eeg_subj = min(19, S // 2)
        subj = torch.where(is_eeg, torch.randint(0, n_eeg_subj, (B,), generator=gen),
                           torch.randint(n_eeg_subj, S, (B,), generator=gen))
    else:
        subj = torch.randint(0, S, (B,), generator=gen)
    if planted:
        L = 32
        z = torch.randn(B, L, T + 7, generator=gen)
        z = z.unfold(2, 8, 1).mean(-1)                      # 8-tap moving average -> [B, L, T]
        wgen = torch.Generator().manual_seed(world_seed)
        A = torch.randn(S, C, L, generator=wgen) / L ** 0.5
        W = torch.randn(F, L, generator=wgen)
        meg = torch.einsum("bcl,blt->bct", A[subj], z) + noise * torch.randn(B, C, T, generator=gen)
        feats = torch.einsum("fl,blt->bft", W, torch.roll(z, 18, dims=2)) \
            + noise * torch.randn(B, F, T, generator=gen)
        meg = meg / meg.std()
        feats = (feats - feats.mean()) / feats.std()
    else:
        meg = torch.randn(B, C, T, generator=gen)
        feats = torch.randn(B, F, T, generator=gen)
    meg = meg.clamp(-20, 20)
    recs = [recordings[int(i)] for i in rec_idx]
    for i, rec in enumerate(recs):                          # padded sensors carry zeros
        meg[i, len(rec.layout):] = 0
    mask = torch.ones(B, 1, T, dtype=torch.bool)
    return SegmentBatch(meg, feats, mask, subj, rec_idx.clone(), recs)


def make_config_batch(name: str, seed: int = 2036, batch: tp.Optional[int] = None,
                      **kw) -> SegmentBatch:
    c = dict(CONFIGS[name])
    if batch is not None:
        c["B"] = batch
    if name == "cfg1":
        kw.setdefault("n_layouts", 4)      # studies/fake.py:124: 4 fake recordings
    if name == "cfg5":
        kw.setdefault("mixed_eeg", True)
    return make_batch(c["B"], c["C"], c["T"], c["F"], c["S"], seed=seed, **kw)
