"""Generate golden vectors by running the REAL reference code (read-only /root/reference).

Run in the build container only:   python tests/golden/make_golden.py
Outputs small ``tests/golden/*.npz`` fixtures (committed).  Each fixture holds: the seeded
inputs, the reference ``state_dict`` before the step, the reference forward output, ClipLoss
value, every parameter gradient, and the ``state_dict`` after ``n_steps`` of
``zero_grad -> backward -> Adam.step`` (bm/solver.py:384-387 with the optimizer of
bm/train.py:118-119).

The reference has no golden vectors for this path; these fixtures are the pin
for ``oracle/bm_oracle.py`` and, through it (and directly), for the HIP path.
"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))

from _ref_import import load_reference, load_reference_norm  # noqa: E402
from brainmagick_amd.synthetic import make_batch  # noqa: E402

SMALL = dict(C=20, T=48, F=10, S=5, B=6)

# name -> (SimpleConv kwargs, hidden, training, extra)
BASE = dict(depth=10, kernel_size=3, dilation_growth=2, dilation_period=5, batch_norm=True,
            skip=True, gelu=True, glu=2, glu_context=1, glu_glu=True, complex_out=True,
            merger=True, merger_pos_dim=32, merger_channels=12, merger_dropout=0.2,
            initial_linear=12, initial_depth=1, subject_layers=True, subject_layers_dim="input",
            subject_dim=0)


def variant(**kw):
    out = dict(BASE)
    out.update(kw)
    return out


VARIANTS = {
    # the paper model, scaled down; train mode with an injected ban centre; padded sensors
    "clip_conv_train": dict(cfg=BASE, training=True, n_steps=2, n_layouts=2, pad_layout=True),
    # eval mode: running BN stats, no sensor dropout
    "clip_conv_eval": dict(cfg=BASE, training=False, n_steps=0, n_layouts=2, pad_layout=True),
    # ablations from bm/grids/nmi/ablation_final.py:42-50
    "no_merger_relu_noskip": dict(cfg=variant(merger=False, gelu=False, skip=False, glu=0,
                                              initial_linear=0), training=True, n_steps=1),
    "no_subject_layers_leaky": dict(cfg=variant(subject_layers=False, gelu=False,
                                                relu_leakiness=0.1, merger_dropout=0.0),
                                    training=True, n_steps=1),
    "subject_embedding": dict(cfg=variant(subject_layers=False, subject_dim=8),
                              training=True, n_steps=1),
    "plain_out": dict(cfg=variant(complex_out=False), training=True, n_steps=1),
    "linear_out_k5": dict(cfg=variant(complex_out=False, linear_out=True, kernel_size=5,
                                      depth=4, dilation_period=None, glu=1, glu_context=0,
                                      batch_norm=False),
                          training=True, n_steps=1),
    "initial_depth2_hidden_subject": dict(
        cfg=variant(initial_depth=2, initial_nonlin=True, subject_layers_dim="hidden"),
        training=True, n_steps=1),
    "subsample_channels": dict(cfg=variant(subsample_meg_channels=9), training=True, n_steps=1),
    # more candidates than estimates (negatives appended, bm/solver.py:359-371)
    "extra_negatives": dict(cfg=BASE, training=True, n_steps=1, extra_negatives=7),
    # options outside the paper's grids (round 4: implemented off the hot path instead of rejected): LayerScale,
    # rewrite conv, post-skip depthwise conv, the (constant) merger usage penalty -- train mode, one step
    "layer_scale_rewrite_post_skip": dict(cfg=variant(depth=4, scale=0.5, rewrite=True, post_skip=True,
                                                      relu_leakiness=0.1, merger_penalty=0.3),
                                          training=True, n_steps=1, n_layouts=2),
    # ChannelDropout with every random centre pinned to the injected one (fake_rand), rescaled
    "channel_dropout_train": dict(cfg=variant(depth=4, dropout=0.35, dropout_rescale=True), training=True, n_steps=1,
                                  n_layouts=2, pad_layout=True),
    # the Dropout modules shift the sequence indices (state_dict keys); eval mode: identity
    "conv_dropouts_eval": dict(cfg=variant(depth=4, conv_dropout=0.2, dropout_input=0.1, dropout=0.3),
                               training=False, n_steps=0, n_layouts=2, pad_layout=True),
    # grouped convs from the second layer on (common.py:113-114)
    "groups2": dict(cfg=variant(depth=4, groups=2), training=True, n_steps=1, n_layouts=2),
    # one set of attention heads per subject
    "merger_per_subject": dict(cfg=variant(depth=4, merger_per_subject=True), training=True, n_steps=1, n_layouts=2,
                               pad_layout=True),
    # a second input next to the sensors: one conv stack per input, outputs side by side into the head
    # (simpleconv.py:149-176,228-234) ...
    "two_inputs": dict(cfg=variant(depth=4), training=True, n_steps=1, n_layouts=2,
                       extra_inputs={"aux": 6}, extra_hidden={"aux": 8}),
    # ... or the inputs side by side into one stack (`concatenate`, simpleconv.py:143-147,223-226)
    "two_inputs_concatenate": dict(cfg=variant(depth=4, concatenate=True, complex_out=False, linear_out=True),
                                   training=True, n_steps=1, n_layouts=2,
                                   extra_inputs={"aux": 6}, extra_hidden={"aux": 8}),
    # the LSTM stack between the conv stack and the head (simpleconv.py:163-165,236-237; T = 48 is padded to 50)
    "dual_path": dict(cfg=variant(depth=4, dual_path=1), training=True, n_steps=1, n_layouts=2),
}
# fixtures the CPU oracle (oracle/bm_oracle.py restates the hot path only) does not cover: HIP path vs reference directly
OFF_PATH = ("layer_scale_rewrite_post_skip", "channel_dropout_train", "conv_dropouts_eval", "merger_per_subject",
            "groups2", "two_inputs", "two_inputs_concatenate", "dual_path")
HIDDEN = 16


class _Batch:
    def __init__(self, sb):
        self.meg = sb.meg
        self.subject_index = sb.subject_index
        self._recordings = sb._recordings
        self._positions = sb.positions()

    def __len__(self):
        return len(self.meg)


def run_variant(name, spec, sc, common, losses):
    cfg = dict(spec["cfg"])
    training = spec["training"]
    seed = 2036 + sum(map(ord, name))
    sb = make_batch(SMALL["B"], SMALL["C"], SMALL["T"], SMALL["F"], SMALL["S"], seed=seed,
                    n_layouts=spec.get("n_layouts", 1))
    if spec.get("pad_layout"):
        # second recording has only 15 valid sensors (padded ones are INVALID / zero meg)
        sb._recordings = [r if r.recording_index == 0 else
                          type(r)(r.recording_index, r.layout[:15], r.study)
                          for r in sb._recordings]
        for i, r in enumerate(sb._recordings):
            sb.meg[i, len(r.layout):] = 0
    gen = torch.Generator().manual_seed(seed + 1)
    n_neg = spec.get("extra_negatives", 0)
    candidates = sb.features
    if n_neg:
        candidates = torch.cat([candidates, torch.randn(n_neg, SMALL["F"], SMALL["T"],
                                                        generator=gen)])
    ban_center = torch.rand(2, generator=gen)
    extra_inputs = {k: torch.randn(SMALL["B"], ch, SMALL["T"], generator=gen)
                    for k, ch in spec.get("extra_inputs", {}).items()}

    torch.manual_seed(seed)
    model = sc.SimpleConv(in_channels={"meg": SMALL["C"], **spec.get("extra_inputs", {})}, out_channels=SMALL["F"],
                          hidden={"meg": HIDDEN, **spec.get("extra_hidden", {})}, n_subjects=SMALL["S"], **cfg)
    # non-trivial BN affine / running stats so that eval mode and BN grads are exercised
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.uniform_(0.5, 1.5, generator=gen)
                mod.bias.uniform_(-0.3, 0.3, generator=gen)
                mod.running_mean.uniform_(-0.2, 0.2, generator=gen)
                mod.running_var.uniform_(0.5, 1.5, generator=gen)
    loss_mod = losses.ClipLoss()
    optim = torch.optim.Adam(model.parameters(), lr=3e-4, betas=(0.9, 0.999))
    model.train(training)
    loss_mod.train(training)

    batch = _Batch(sb)
    common.PositionGetter.get_positions = lambda self, b: b._positions.clone()
    real_rand = torch.rand

    def fake_rand(*a, **k):          # bm/models/common.py:343 `torch.rand(2, device=...)`
        if a == (2,):
            return ban_center.clone()
        return real_rand(*a, **k)

    out = {"meta": json.dumps(dict(cfg=cfg, hidden=HIDDEN, training=training, **SMALL,
                                   n_steps=spec["n_steps"], torch=torch.__version__,
                                   extra_inputs=spec.get("extra_inputs", {}),
                                   extra_hidden=spec.get("extra_hidden", {})))}
    for k, v in extra_inputs.items():
        out[f"in/{k}"] = v.numpy().copy()
    for k, v in model.state_dict().items():
        out[f"sd0/{k}"] = v.detach().clone().numpy()
    out["in/meg"] = sb.meg.numpy().copy()
    out["in/positions"] = batch._positions.numpy().copy()
    out["in/subjects"] = sb.subject_index.numpy().copy()
    out["in/candidates"] = candidates.numpy().copy()
    out["in/ban_center"] = ban_center.numpy().copy()

    common.torch.rand = fake_rand
    try:
        losses_seen = []
        steps = max(spec["n_steps"], 1)
        for step in range(steps):
            inputs = {"meg": sb.meg.clone(), **{k: v.clone() for k, v in extra_inputs.items()}}
            mask = torch.ones(len(sb.meg), 1, SMALL["T"], dtype=torch.bool)
            if training and spec["n_steps"]:
                estimate = model(inputs, batch)
                loss = loss_mod(estimate, candidates, mask)
                optim.zero_grad()
                loss.backward()
                if step == 0:
                    out["out/estimate"] = estimate.detach().numpy().copy()
                    for k, p in model.named_parameters():
                        out[f"grad/{k}"] = p.grad.detach().numpy().copy()
                optim.step()
            else:
                with torch.no_grad():
                    estimate = model(inputs, batch)
                    loss = loss_mod(estimate, candidates, mask)
                    out["out/estimate"] = estimate.numpy().copy()
                    out["out/probabilities"] = loss_mod.get_probabilities(
                        estimate, candidates).numpy().copy()
            losses_seen.append(float(loss))
    finally:
        common.torch.rand = real_rand
    out["out/losses"] = np.asarray(losses_seen, dtype=np.float64)
    if name in OFF_PATH and getattr(model, "merger", None) is not None:
        out["out/training_penalty"] = np.asarray(float(model.merger.training_penalty))
    for k, v in model.state_dict().items():
        out[f"sd1/{k}"] = v.detach().clone().numpy()
    return out


def clip_only_fixture(losses):
    """ClipLoss options (pool / center / trim) on their own, incl. B' > B."""
    gen = torch.Generator().manual_seed(77)
    est = torch.randn(5, 7, 33, generator=gen)
    cand = torch.randn(9, 7, 33, generator=gen) * 3 + 0.5
    out = {"in/estimate": est.numpy().copy(), "in/candidate": cand.numpy().copy()}

    class DsetArgs:
        tmin = -0.5
        sample_rate = 20

    for tag, kw in [("plain", {}), ("pool", dict(pool=True)), ("center", dict(center=True)),
                    ("trim", dict(tmin=-0.2, tmax=0.9, dset_args=DsetArgs()))]:
        mod = losses.ClipLoss(**kw)
        mod.eval()
        e = est.clone().requires_grad_(True)
        loss = mod(e, cand, torch.ones(5, 1, 33, dtype=torch.bool))
        loss.backward()
        out[f"{tag}/scores"] = mod.get_scores(est, cand).numpy().copy()
        out[f"{tag}/probabilities"] = mod.get_probabilities(est, cand).numpy().copy()
        out[f"{tag}/loss"] = np.asarray(float(loss))
        out[f"{tag}/grad_estimate"] = e.grad.numpy().copy()
    return out


def scale_reject_fixture():
    """bm/norm.py BatchScaler._transform + ScaleReject.__call__ of the live reference."""
    import collections
    norm = load_reference_norm()
    gen = torch.Generator().manual_seed(4242)
    B, C, T, Fd, R = 10, 14, 40, 6, 3
    meg = torch.randn(B, C, T, generator=gen) * 4 + 0.3
    meg[1, 3, 7] = 400.0
    meg[6, 0, 0] = -250.0
    features = torch.randn(B, Fd, T, generator=gen) * 2 + 1
    rec = torch.randint(0, R, (B,), generator=gen)

    class FB:                                   # the FeaturesBuilder surface _transform touches
        dimension = Fd

        def get_slice(self, name):
            return {"a": slice(0, 4), "b": slice(4, 6)}[name]

        def items(self):
            return []

    class Batch:
        def __init__(self, **kw):
            self.__dict__.update(kw)

        def replace(self, **kw):
            d = dict(self.__dict__)
            d.update(kw)
            return Batch(**d)

        def __getitem__(self, keep):
            return Batch(meg=self.meg[keep], features=self.features[keep],
                         features_mask=self.features_mask[keep],
                         recording_index=self.recording_index[keep])

    scaler = norm.BatchScaler.__new__(norm.BatchScaler)
    scaler.features_builder = FB()
    scaler.meg_scalers = {}
    centers, scales = torch.zeros(R, C), torch.ones(R, C)
    for r in range(R):
        rs = norm.RobustScaler()
        rs.fit(torch.randn(500, C, generator=gen) * (r + 1) + 0.1 * r)
        scaler.meg_scalers[r] = rs
        centers[r], scales[r] = rs.center_, rs.scale_
    fa, fb = norm.StandardScaler(per_channel=True), norm.NoOpScaler()
    fa.fit(torch.randn(300, 4, generator=gen) * 2 + 1, torch.ones(300, 1, dtype=torch.bool))
    scaler.feature_scalers = collections.OrderedDict(a=fa, b=fb)
    out = {"in/meg": meg.numpy().copy(), "in/features": features.numpy().copy(),
           "in/recording_index": rec.numpy().copy(), "in/meg_center": centers.numpy().copy(),
           "in/meg_scale": scales.numpy().copy(),
           "in/feature_center": torch.cat([fa.center_, torch.zeros(2)]).numpy().copy(),
           "in/feature_scale": torch.cat([fa.scale_, torch.ones(2)]).numpy().copy()}
    for tag, clip in (("clip", True), ("reject", False)):
        batch = Batch(meg=meg.clone(), features=features.clone(),
                      features_mask=torch.ones(B, 1, T, dtype=torch.bool), recording_index=rec)
        sr = norm.ScaleReject(scaler, limit=20, clip=clip)
        res, keep = sr(batch)
        out[f"{tag}/meg"] = res.meg.numpy().copy()
        out[f"{tag}/features"] = res.features.numpy().copy()
        out[f"{tag}/keep"] = keep.numpy().copy()
    return out



def _extract_function(path, name):
    """Source of the top-level function `name` of a reference file, taken from its AST (the module itself cannot
    be imported: it pulls flashy / dora / mne)."""
    import ast
    text = Path(path).read_text()
    for node in ast.parse(text).body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            return ast.get_source_segment(text, node), node, text
    raise KeyError(name)


def retrieval_fixture(losses):
    """The retrieval rules as the reference computes them, executed from the reference sources:
      * scripts/run_eval_probs.py:237-264 `_get_accuracy_from_probs` (a pure function: compiled as is);
      * bm/wer.py:82-121, the per-segment loop of `get_wer` from `negatives = negatives.to(solver.device)` to the
        metrics dict, run statement by statement under a stub `solver` (device "cpu", the reference ClipLoss,
        `args.test.wer_topx`), with `LogProgress` = identity."""
    import ast
    import types
    REF = Path("/root/reference")
    gen = torch.Generator().manual_seed(77)
    # --- segment / vocabulary accuracy -----------------------------------------------------------------
    src, _, _ = _extract_function(REF / "scripts" / "run_eval_probs.py", "_get_accuracy_from_probs")
    ns = {"torch": torch}
    exec(compile(src, "run_eval_probs._get_accuracy_from_probs", "exec"), ns)
    acc_fn = ns["_get_accuracy_from_probs"]
    Bq, V = 96, 150
    probs = torch.softmax(2.0 * torch.randn(Bq, V, generator=gen), dim=1)
    vocab_labels = torch.randint(0, 60, (V,), generator=gen)           # several columns share a label
    target_labels = vocab_labels[torch.randint(0, V, (Bq,), generator=gen)].clone()
    target_labels[:7] = 1000                                             # labels no column carries
    for b in range(Bq):                                                  # make the targets partly retrievable
        if b % 3 == 0:
            cols = (vocab_labels == target_labels[b]).nonzero().flatten()
            if len(cols):
                probs[b, cols[0]] += 0.05
    probs = probs / probs.sum(1, keepdim=True)
    out = {"acc/probs": probs.numpy().copy(), "acc/vocab_labels": vocab_labels.numpy().copy(),
           "acc/target_labels": target_labels.numpy().copy()}
    for k in (1, 5, 10):
        out[f"acc/top{k}"] = np.array(acc_fn(probs, target_labels, vocab_labels, topk=k))
    # --- word-level WER loop -----------------------------------------------------------------------------
    N, Fd, T, n_neg, topx, seed = 60, 6, 20, 32, 3, 11
    estimates = torch.randn(N, Fd, T, generator=gen)
    outputs = 0.12 * estimates + torch.randn(N, Fd, T, generator=gen)    # estimates carry SOME signal about their target
    word_hashes = torch.randint(1, 19, (N,), generator=gen).int()        # 18 words: duplicates among negatives
    kept = torch.randperm(N, generator=torch.Generator().manual_seed(seed))[:n_neg]
    _, fn_node, text = _extract_function(REF / "bm" / "wer.py", "get_wer")
    body = fn_node.body
    first = next(i for i, st in enumerate(body)
                 if isinstance(st, ast.Assign) and ast.get_source_segment(text, st).startswith("negatives = negatives.to("))
    last = next(i for i, st in enumerate(body)
                if isinstance(st, ast.Assign) and ast.get_source_segment(text, st).startswith("metrics ="))
    import textwrap
    loop_src = "\n".join(textwrap.dedent(" " * st.col_offset + ast.get_source_segment(text, st))
                         for st in body[first:last + 1])
    solver = types.SimpleNamespace(
        device="cpu", loss=losses.ClipLoss(),
        args=types.SimpleNamespace(num_prints=1, test=types.SimpleNamespace(wer_topx=topx, wer_random=False)))
    env = {"torch": torch, "solver": solver, "ClipLoss": losses.ClipLoss, "logger": None,
           "LogProgress": lambda logger, it, **kw: it, "estimates": estimates, "outputs": outputs,
           "word_hashes": word_hashes, "negatives": outputs[kept], "negative_hashes": word_hashes[kept]}
    exec(compile(loop_src, "bm/wer.py:get_wer[loop]", "exec"), env)
    out.update({"wer/estimates": estimates.numpy().copy(), "wer/outputs": outputs.numpy().copy(),
                "wer/word_hashes": word_hashes.numpy().copy(), "wer/kept": kept.numpy().copy(),
                "wer/wer": np.array(env["metrics"]["wer"]), "wer/wer_vocab": np.array(env["metrics"]["wer_vocab"]),
                "wer/meta": np.array(json.dumps(dict(n_negatives=n_neg, topx=topx, perm_seed=seed)))})
    return out


def mne_layout_fixture(common):
    """bm/models/common.py:190-222 `PositionGetter.get_recording_layout`, the branch real recordings take, under
    a stub `mne.find_layout` that returns a layout object of the shape mne's has (`names`, `pos` [n, 4] float64):
    channel names with the '-<id>' suffix of CTF/KIT files, two channels the layout does not know, layout rows in
    a different order than the recording's channels."""
    import types
    gen = np.random.default_rng(5)
    n_layout = 30
    names = [f"MEG {k:03d}" for k in range(n_layout)]
    order = gen.permutation(n_layout)
    layout = types.SimpleNamespace(names=[names[k] for k in order],
                                   pos=np.concatenate([gen.uniform(-0.4, 0.55, (n_layout, 2)),
                                                       np.full((n_layout, 2), 0.04)], axis=1))
    ch_names = [f"MEG {k:03d}-4507" for k in range(26)] + ["UADC001-4507", "MEG 027", "STIM-1", "MEG 029-12"]
    info = types.SimpleNamespace(ch_names=ch_names)
    common.mne.find_layout = lambda _info: layout
    rec = types.SimpleNamespace(recording_index=3, mne_info=info, study_name=lambda: "stub", recording_uid="stub_0")
    positions = common.PositionGetter().get_recording_layout(rec)
    return {"layout_names": np.array(layout.names), "layout_pos": layout.pos.copy(),
            "ch_names": np.array(ch_names), "positions": positions.numpy().copy()}


def wide_kernels_fixture(sc, common, losses):
    """The paper architecture at the smallest size the headline wide f16x2 kernels take (hidden 256, depth 4, 64
    sensors, T = 192, batch 8), two training steps of the live reference.  Parameters (1.6 M) and inputs are rebuilt
    from the seed by the consumer (tests/helpers.py: wide_inputs, randomize_batchnorm); stored: digests of both, the
    forward output, the losses of both steps, and per parameter the gradient norm, 64 sampled gradient elements and the
    same 64 elements of the parameter after the two Adam steps."""
    sys.path.insert(0, str(HERE.parent))
    import helpers as Hh
    d = Hh.WIDE_DIMS
    sb, candidates, ban_center, gen = Hh.wide_inputs()
    torch.manual_seed(d["seed"])
    model = sc.SimpleConv(in_channels={"meg": d["C"]}, out_channels=d["F"], hidden={"meg": d["hidden"]},
                          n_subjects=d["S"], **Hh.WIDE_CFG)
    Hh.randomize_batchnorm(model, gen)
    # the replacement classes must rebuild the very same parameters from the seed
    from brainmagick_amd.models import SimpleConv as HipSimpleConv
    torch.manual_seed(d["seed"])
    gen2 = Hh.wide_inputs()[3]
    mine = HipSimpleConv(in_channels={"meg": d["C"]}, out_channels=d["F"], hidden={"meg": d["hidden"]},
                         n_subjects=d["S"], **Hh.WIDE_CFG)
    Hh.randomize_batchnorm(mine, gen2)
    sd_ref, sd_mine = model.state_dict(), mine.state_dict()
    assert list(sd_ref) == list(sd_mine)
    for k in sd_ref:
        assert torch.equal(sd_ref[k], sd_mine[k]), k
    loss_mod = losses.ClipLoss()
    optim = torch.optim.Adam(model.parameters(), lr=3e-4, betas=(0.9, 0.999))
    model.train(True)
    loss_mod.train(True)
    batch = _Batch(sb)
    common.PositionGetter.get_positions = lambda self, b: b._positions.clone()
    real_rand = torch.rand

    def fake_rand(*a, **k):
        if a == (2,):
            return ban_center.clone()
        return real_rand(*a, **k)

    out = {"meta": json.dumps(dict(dims=d, cfg=Hh.WIDE_CFG, n_steps=2, torch=torch.__version__,
                                   n_params=sum(p.numel() for p in model.parameters())))}
    for k, v in model.state_dict().items():
        out[f"sd0_digest/{k}"] = Hh.tensor_digest(v)
    out["in_digest/meg"] = Hh.tensor_digest(sb.meg)
    out["in_digest/candidates"] = Hh.tensor_digest(candidates)
    out["in_digest/positions"] = Hh.tensor_digest(batch._positions)
    out["in/ban_center"] = ban_center.numpy().copy()
    common.torch.rand = fake_rand
    try:
        seen = []
        for step in range(2):
            mask = torch.ones(len(sb.meg), 1, d["T"], dtype=torch.bool)
            estimate = model({"meg": sb.meg.clone()}, batch)
            loss = loss_mod(estimate, candidates, mask)
            optim.zero_grad()
            loss.backward()
            if step == 0:
                out["out/estimate"] = estimate.detach().numpy().copy()
                for k, p in model.named_parameters():
                    g = p.grad.detach().flatten()
                    idx = Hh.sample_indices(g.numel())
                    out[f"grad_norm/{k}"] = np.array(float(g.double().norm()))
                    out[f"grad_max/{k}"] = np.array(float(g.abs().max()))
                    out[f"grad_sample/{k}"] = g[idx].numpy().copy()
            optim.step()
            seen.append(float(loss))
    finally:
        common.torch.rand = real_rand
    out["out/losses"] = np.asarray(seen, dtype=np.float64)
    for k, p in model.named_parameters():
        out[f"sd1_sample/{k}"] = p.detach().flatten()[Hh.sample_indices(p.numel())].numpy().copy()
    return out


def main(only=()):
    """`python make_golden.py` regenerates everything; `python make_golden.py <variant> ...` only those models."""
    sc, common, losses = load_reference()
    only = tuple(only)
    if only == ("wide_kernels_train",) or not only:
        out = wide_kernels_fixture(sc, common, losses)
        np.savez_compressed(HERE / "wide_kernels_train.npz", **out)
        print(f"wide_kernels_train: losses={out['out/losses']}")
        if only:
            return
    if only:
        for name in only:
            out = run_variant(name, VARIANTS[name], sc, common, losses)
            np.savez_compressed(HERE / f"{name}.npz", **out)
            print(f"{name}: losses={out['out/losses']}")
        return
    np.savez_compressed(HERE / "scale_reject.npz", **scale_reject_fixture())
    print("scale_reject: done")
    for name, spec in VARIANTS.items():
        out = run_variant(name, spec, sc, common, losses)
        np.savez_compressed(HERE / f"{name}.npz", **out)
        print(f"{name}: losses={out['out/losses']}")
    np.savez_compressed(HERE / "clip_loss.npz", **clip_only_fixture(losses))
    print("clip_loss: done")
    np.savez_compressed(HERE / "retrieval_rules.npz", **retrieval_fixture(losses))
    print("retrieval_rules: done")
    np.savez_compressed(HERE / "mne_layout.npz", **mne_layout_fixture(common))
    print("mne_layout: done")


if __name__ == "__main__":
    main(sys.argv[1:])
