"""Import the *real* reference modules from /root/reference (read-only) for golden generation.

Only used by tests/golden/make_golden.py in the build container. Nothing that runs on the
GPU box imports this file (``/root/reference`` does not exist there).

Recipe: ``bm/losses.py`` needs only torch; ``bm/models/{common,simpleconv}.py``
need stub modules for ``mne``, ``torchaudio`` and a skeleton ``bm`` / ``bm.studies.api`` package.
"""
import importlib
import importlib.util
import sys
import types
from pathlib import Path

REF = Path("/root/reference")


def load_reference():
    """Returns (simpleconv_module, common_module, losses_module) of the reference."""
    if not REF.exists():
        raise RuntimeError("/root/reference not available: golden vectors can only be "
                           "regenerated in the build container")
    if "bm.models.simpleconv" not in sys.modules:
        mne = types.ModuleType("mne")
        ta = types.ModuleType("torchaudio")
        ta.transforms = types.SimpleNamespace(Spectrogram=None)
        sys.modules.setdefault("mne", mne)
        sys.modules.setdefault("torchaudio", ta)
        bm = types.ModuleType("bm")
        bm.__path__ = []  # namespace skeleton: do NOT run bm/__init__.py (pulls mne, dora, ...)
        studies = types.ModuleType("bm.studies")
        studies.__path__ = []
        api = types.ModuleType("bm.studies.api")
        api.Recording = object
        models = types.ModuleType("bm.models")
        models.__path__ = [str(REF / "bm" / "models")]
        sys.modules.update({"bm": bm, "bm.studies": studies, "bm.studies.api": api,
                            "bm.models": models})
    simpleconv = importlib.import_module("bm.models.simpleconv")
    common = importlib.import_module("bm.models.common")
    if "bm_ref_losses" not in sys.modules:
        spec = importlib.util.spec_from_file_location("bm_ref_losses", REF / "bm" / "losses.py")
        losses = importlib.util.module_from_spec(spec)
        sys.modules["bm_ref_losses"] = losses
        spec.loader.exec_module(losses)
    losses = sys.modules["bm_ref_losses"]
    return simpleconv, common, losses


def load_reference_norm():
    """bm/norm.py under stubs for its three imports (dora.log, .features, .dataset)."""
    load_reference()
    if "bm.norm" not in sys.modules:
        dora = types.ModuleType("dora")
        dora_log = types.ModuleType("dora.log")
        dora_log.LogProgress = object
        dora.log = dora_log
        sys.modules.setdefault("dora", dora)
        sys.modules.setdefault("dora.log", dora_log)
        feats = types.ModuleType("bm.features")
        feats.FeaturesBuilder = object
        feats.Feature = object
        dset = types.ModuleType("bm.dataset")
        dset.SegmentBatch = object
        sys.modules["bm.features"] = feats
        sys.modules["bm.dataset"] = dset
        spec = importlib.util.spec_from_file_location("bm.norm", REF / "bm" / "norm.py")
        mod = importlib.util.module_from_spec(spec)
        sys.modules["bm.norm"] = mod
        spec.loader.exec_module(mod)
    return sys.modules["bm.norm"]
