"""oracle/bm_oracle.py vs golden vectors produced by the live reference (CPU only)."""
import pytest
import torch

from oracle import bm_oracle as O
from helpers import Golden, MODEL_FIXTURES, rel_l2

# fp32 oracle vs fp32 reference: same torch kernels in (nearly) the same order
FWD_TOL = 2e-6
GRAD_TOL = 2e-5


@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_oracle_matches_reference(name):
    g = Golden(name)
    meta = g.meta
    sd0, sd1, grads = g.group("sd0"), g.group("sd1"), g.group("grad")
    inp = g.group("in")
    model = O.OracleModel(sd0, meta["cfg"], meta["hidden"], meta["F"])
    training = meta["training"]
    if training and meta["n_steps"]:
        losses = []
        for step in range(meta["n_steps"]):
            loss, est, gr = model.train_step(inp["meg"], inp["positions"], inp["subjects"],
                                             inp["candidates"], inp["ban_center"])
            losses.append(float(loss))
            if step == 0:
                assert rel_l2(est, g.t("out/estimate")) < FWD_TOL
                assert set(gr) == set(grads)
                for k in grads:
                    assert rel_l2(gr[k], grads[k]) < GRAD_TOL, k
        ref_losses = g.raw["out/losses"]
        assert max(abs(a - b) for a, b in zip(losses, ref_losses)) < 1e-5
        for k, v in sd1.items():
            if v.is_floating_point():
                assert rel_l2(model.sd[k], v) < 1e-5, k
            else:
                assert int(model.sd[k]) == int(v), k
    else:
        est = model.forward(inp["meg"], inp["positions"], inp["subjects"], training=False)
        assert rel_l2(est, g.t("out/estimate")) < FWD_TOL
        loss = O.clip_loss(est, inp["candidates"])
        assert abs(float(loss) - g.raw["out/losses"][0]) < 1e-5
        probs = O.clip_probabilities(est, inp["candidates"])
        assert rel_l2(probs, g.t("out/probabilities")) < 1e-5


def _wide_setup():
    """Inputs and initial state of the wide-kernel fixture, rebuilt from the seed; the fixture's digests prove it."""
    import helpers as Hh
    from brainmagick_amd.models import SimpleConv
    g = Golden("wide_kernels_train")
    d = Hh.WIDE_DIMS
    sb, candidates, ban_center, gen = Hh.wide_inputs()
    torch.manual_seed(d["seed"])
    model = SimpleConv(in_channels={"meg": d["C"]}, out_channels=d["F"], hidden={"meg": d["hidden"]},
                       n_subjects=d["S"], **Hh.WIDE_CFG)
    Hh.randomize_batchnorm(model, gen)
    for k, v in model.state_dict().items():
        assert (Hh.tensor_digest(v) == g.raw[f"sd0_digest/{k}"]).all(), f"initial state differs from the fixture's: {k}"
    assert (Hh.tensor_digest(sb.meg) == g.raw["in_digest/meg"]).all()
    assert (Hh.tensor_digest(candidates) == g.raw["in_digest/candidates"]).all()
    assert (Hh.tensor_digest(sb.positions()) == g.raw["in_digest/positions"]).all()
    assert torch.equal(ban_center, g.t("in/ban_center"))
    return g, sb, candidates, ban_center, model


def test_oracle_matches_reference_at_the_wide_kernel_shape():
    """The oracle against the live reference's two training steps of the smallest model the wide f16x2 kernels cover
    (hidden 256, depth 4, 64 sensors, T = 192; tests/golden/make_golden.py: wide_kernels_fixture)."""
    import helpers as Hh
    g, sb, candidates, ban_center, model = _wide_setup()
    d = Hh.WIDE_DIMS
    oracle = O.OracleModel({k: v.clone() for k, v in model.state_dict().items()}, Hh.WIDE_CFG, d["hidden"], d["F"])
    losses = []
    for step in range(2):
        loss, est, gr = oracle.train_step(sb.meg, sb.positions(), sb.subject_index, candidates, ban_center)
        losses.append(float(loss))
        if step == 0:
            assert rel_l2(est, g.t("out/estimate")) < FWD_TOL
            gscale = max(float(g.raw[k]) for k in g.raw if k.startswith("grad_norm/"))
            for k, v in gr.items():
                ref_norm = float(g.raw[f"grad_norm/{k}"])
                if float(g.raw[f"grad_max/{k}"]) <= 1e-5 * gscale:
                    continue                      # round-off noise in the reference itself (conv bias before BatchNorm)
                assert abs(float(v.double().norm()) - ref_norm) < GRAD_TOL * ref_norm, k
                idx = Hh.sample_indices(v.numel())
                assert (v.flatten()[idx].double() - g.t(f"grad_sample/{k}").double()).norm() < 5 * GRAD_TOL * ref_norm, k
    assert max(abs(a - b) for a, b in zip(losses, g.raw["out/losses"])) < 1e-5


def test_oracle_fp64_close_to_fp32_reference():
    g = Golden("clip_conv_eval")
    model = O.OracleModel(g.group("sd0"), g.meta["cfg"], g.meta["hidden"], g.meta["F"],
                          dtype=torch.float64)
    inp = g.group("in")
    est = model.forward(inp["meg"], inp["positions"], inp["subjects"])
    assert rel_l2(est, g.t("out/estimate")) < 1e-5


@pytest.mark.parametrize("tag,kw", [("plain", {}), ("pool", dict(pool=True)),
                                    ("center", dict(center=True)), ("trim", None)])
def test_clip_loss_options(tag, kw):
    g = Golden("clip_loss")
    est, cand = g.t("in/estimate").requires_grad_(True), g.t("in/candidate")
    e, c = est, cand
    if kw is None:
        e, c = O.clip_trim(est, cand, tmin=-0.2, tmax=0.9, dset_tmin=-0.5, sample_rate=20)
        kw = {}
    scores = O.clip_scores(e, c, **kw)
    assert rel_l2(scores, g.t(f"{tag}/scores")) < 1e-6
    loss = O.clip_loss(e, c, **kw)
    assert abs(float(loss) - float(g.raw[f"{tag}/loss"])) < 1e-6
    loss.backward()
    assert rel_l2(est.grad, g.t(f"{tag}/grad_estimate")) < 1e-5
    assert rel_l2(O.clip_probabilities(e, c, **kw), g.t(f"{tag}/probabilities")) < 1e-6


def test_invalid_sensors_have_no_influence():
    """INVALID-position sensors carry exactly zero attention weight."""
    g = Golden("clip_conv_eval")
    inp = g.group("in")
    model = O.OracleModel(g.group("sd0"), g.meta["cfg"], g.meta["hidden"], g.meta["F"])
    base = model.forward(inp["meg"], inp["positions"], inp["subjects"])
    meg = inp["meg"].clone()
    invalid = O.is_invalid(inp["positions"])
    assert invalid.any()
    meg[invalid] = 123.0
    assert torch.equal(model.forward(meg, inp["positions"], inp["subjects"]), base)


def test_topk_accuracy_rule():
    probs = torch.tensor([[0.1, 0.7, 0.2], [0.5, 0.3, 0.2], [0.2, 0.3, 0.5]])
    labels = torch.tensor([0, 1, 2])
    assert O.topk_accuracy(probs, labels, labels, topk=1) == pytest.approx(1 / 3)
    assert O.topk_accuracy(probs, labels, labels, topk=2) == pytest.approx(2 / 3)


@pytest.mark.parametrize("tag,clip", [("clip", True), ("reject", False)])
def test_scale_reject_oracle_matches_reference(tag, clip):
    """oracle.scale_reject vs the live reference's BatchScaler._transform + ScaleReject."""
    g = Golden("scale_reject")
    meg, feats, keep = O.scale_reject(g.t("in/meg"), g.t("in/features"), g.t("in/recording_index"),
                                      g.t("in/meg_center"), g.t("in/meg_scale"),
                                      g.t("in/feature_center"), g.t("in/feature_scale"), limit=20,
                                      clip=clip)
    assert torch.equal(keep, g.t(f"{tag}/keep"))
    assert torch.equal(meg, g.t(f"{tag}/meg"))
    assert torch.equal(feats, g.t(f"{tag}/features"))
    assert clip or int((~keep).sum()) == 2


def test_retrieval_rules_match_the_reference_functions():
    """tests/golden/retrieval_rules.npz was produced by EXECUTING scripts/run_eval_probs.py:237-264
    (`_get_accuracy_from_probs`) and the per-segment loop of bm/wer.py:82-121 from the reference sources
    (tests/golden/make_golden.py::retrieval_fixture); the oracle's restatements must reproduce them."""
    import json
    g = Golden("retrieval_rules")
    probs, vocab, target = g.t("acc/probs"), g.t("acc/vocab_labels"), g.t("acc/target_labels")
    for k in (1, 5, 10):
        assert O.topk_accuracy(probs, vocab, target, k) == pytest.approx(float(g.raw[f"acc/top{k}"]), abs=1e-9)
    assert 0.0 < float(g.raw["acc/top10"]) < 1.0
    meta = json.loads(str(g.raw["wer/meta"]))
    got = O.get_wer_loop(g.t("wer/estimates"), g.t("wer/outputs"), g.t("wer/word_hashes"), g.t("wer/kept"),
                         topx=meta["topx"])
    assert got["wer"] == pytest.approx(float(g.raw["wer/wer"]), abs=1e-9)
    assert got["wer_vocab"] == pytest.approx(float(g.raw["wer/wer_vocab"]), abs=1e-9)
    assert 0.0 < got["wer_vocab"] < got["wer"] < 1.0


def test_mne_layout_branch_matches_the_reference(monkeypatch):
    """`PositionGetter._layout_from_mne` -- the branch every REAL recording takes -- against positions computed by
    the reference's `get_recording_layout` (bm/models/common.py:190-222) under a stub `mne.find_layout`
    (tests/golden/make_golden.py::mne_layout_fixture): suffixed channel names, channels the layout lacks
    (INVALID = -0.1), layout rows in another order than the recording's channels."""
    import sys
    import types
    import numpy as np
    from brainmagick_amd.models.common import PositionGetter
    g = Golden("mne_layout")
    layout = types.SimpleNamespace(names=[str(n) for n in g.raw["layout_names"]], pos=np.array(g.raw["layout_pos"]))
    stub = types.ModuleType("mne")
    stub.find_layout = lambda info: layout
    monkeypatch.setitem(sys.modules, "mne", stub)
    info = types.SimpleNamespace(ch_names=[str(n) for n in g.raw["ch_names"]])
    rec = types.SimpleNamespace(recording_index=7, mne_info=info, layout=None)
    getter = PositionGetter()
    got = getter.get_recording_layout(rec)
    want = g.t("positions")
    assert got.dtype == torch.float32 and got.shape == want.shape
    assert torch.equal(got, want), (got - want).abs().max()
    assert getter.get_recording_layout(rec) is got                       # cached per recording_index
    assert getter._invalid_names == {"UADC001", "STIM"}
    assert int(getter.is_invalid(got).sum()) == 2


def test_symmetric_restatement_is_the_mean_of_the_two_directions():
    """`O.clip_loss_symmetric` (the checker of the opt-in column term; the reference has none) against plain loops."""
    g = torch.Generator().manual_seed(3)
    B, Bc, off = 5, 9, 2
    est = torch.randn(B, 3, 4, generator=g, dtype=torch.float64)
    cand = torch.randn(Bc, 3, 4, generator=g, dtype=torch.float64)
    s = O.clip_scores(est, cand)
    rows = sum(torch.logsumexp(s[b], 0) - s[b, off + b] for b in range(B)) / B
    cols = sum(torch.logsumexp(s[:, off + j], 0) - s[j, off + j] for j in range(B)) / B
    assert float(O.clip_loss_symmetric(est, cand, target_offset=off)) == pytest.approx(float(0.5 * (rows + cols)), abs=1e-12)
    assert float(O.clip_loss(est, cand)) == pytest.approx(
        float(sum(torch.logsumexp(s[b], 0) - s[b, b] for b in range(B)) / B), abs=1e-12)


def test_symmetric_node_restatement_against_plain_loops():
    """`O.clip_loss_symmetric_node` (checker of the symmetric loss with whole-node negatives on both sides) against
    plain loops: rank r's rows against every candidate, its target candidates' columns against every estimate."""
    g = torch.Generator().manual_seed(5)
    world, B, r = 3, 4, 1
    est = torch.randn(world * B, 3, 5, generator=g, dtype=torch.float64)
    cand = torch.randn(world * B, 3, 5, generator=g, dtype=torch.float64)
    s = O.clip_scores(est, cand)
    own = [r * B + k for k in range(B)]
    rows = sum(torch.logsumexp(s[i], 0) - s[i, i] for i in own) / B
    cols = sum(torch.logsumexp(s[:, j], 0) - s[j, j] for j in own) / B
    assert float(O.clip_loss_symmetric_node(est, cand, r, B)) == pytest.approx(float(0.5 * (rows + cols)), abs=1e-12)
    # one rank: the local symmetric loss
    assert float(O.clip_loss_symmetric_node(est[:B], cand[:B], 0, B)) == pytest.approx(
        float(O.clip_loss_symmetric(est[:B], cand[:B])), abs=1e-12)
