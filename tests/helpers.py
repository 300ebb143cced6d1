"""Shared helpers for the parity tests."""
import json
from pathlib import Path

import numpy as np
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"
MODEL_FIXTURES = ["clip_conv_train", "clip_conv_eval", "no_merger_relu_noskip",
                  "no_subject_layers_leaky", "subject_embedding", "plain_out", "linear_out_k5",
                  "initial_depth2_hidden_subject", "subsample_channels", "extra_negatives"]


# options outside the paper's grids, implemented off the hot path (GPU torch ops + the 1x1 HIP conv): the HIP model is
# held to the live reference directly (the CPU oracle restates the hot path only)
OFF_PATH_FIXTURES = ["layer_scale_rewrite_post_skip", "channel_dropout_train", "conv_dropouts_eval",
                     "merger_per_subject", "groups2", "two_inputs", "two_inputs_concatenate",
                     "dual_path"]


class Golden:
    def __init__(self, name):
        self.name = name
        z = np.load(GOLDEN / f"{name}.npz")
        self.raw = {k: z[k] for k in z.files}
        self.meta = json.loads(str(self.raw["meta"])) if "meta" in self.raw else {}

    def group(self, prefix):
        out = {}
        for k, v in self.raw.items():
            if k.startswith(prefix + "/"):
                t = torch.from_numpy(np.array(v))
                out[k[len(prefix) + 1:]] = t
        return out

    def t(self, key):
        return torch.from_numpy(np.array(self.raw[key]))


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    denom = b.norm().item()
    if denom == 0:
        return a.norm().item()
    return (a - b).norm().item() / denom


# ---------------------------------------------------------------------------------------------------------------
# The "wide kernels" fixture (tests/golden/wide_kernels_train.npz): the smallest model whose convs and weight gradients
# run in the headline wide f16x2 kernels (T > 128, M >= 96), held to the LIVE reference directly.  Its 1.6 M
# parameters and its inputs are not stored: both sides rebuild them from the seed (same constructor RNG order, same
# synthetic batch generator) and the fixture keeps digests to prove that they did.
# ---------------------------------------------------------------------------------------------------------------
WIDE_DIMS = dict(B=8, C=64, T=192, F=32, S=4, hidden=256, seed=4077)
WIDE_CFG = dict(depth=4, kernel_size=3, dilation_growth=2, dilation_period=5, batch_norm=True, skip=True, gelu=True,
                glu=2, glu_context=1, glu_glu=True, complex_out=True, merger=True, merger_pos_dim=288,
                merger_channels=256, merger_dropout=0.2, initial_linear=256, initial_depth=1, subject_layers=True,
                subject_layers_dim="input", subject_dim=0)


def wide_inputs():
    """(synthetic batch, candidates, ban centre, generator positioned behind them) of the wide fixture."""
    from brainmagick_amd.synthetic import make_batch
    d = WIDE_DIMS
    sb = make_batch(d["B"], d["C"], d["T"], d["F"], d["S"], seed=d["seed"], n_layouts=2)
    gen = torch.Generator().manual_seed(d["seed"] + 1)
    ban_center = torch.rand(2, generator=gen)
    return sb, sb.features, ban_center, gen


def randomize_batchnorm(model, gen):
    """Non-trivial BatchNorm affine parameters / running statistics, drawn in module order from `gen`."""
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.uniform_(0.5, 1.5, generator=gen)
                mod.bias.uniform_(-0.3, 0.3, generator=gen)
                mod.running_mean.uniform_(-0.2, 0.2, generator=gen)
                mod.running_var.uniform_(0.5, 1.5, generator=gen)


def tensor_digest(t: torch.Tensor):
    """(wrapping int64 sum, xor, element count) of the fp32 bit patterns: exact integer arithmetic, so the digest does
    not depend on the host's summation order (a floating-point sum differs between the build container and the GPU
    box).  Equal digests of two tensors built by the same code from the same seed mean identical tensors for every
    practical purpose."""
    t = t.detach().cpu().contiguous()
    if not t.is_floating_point():
        t = t.float()
    bits = t.float().view(torch.int32).flatten().numpy()
    total = int(bits.astype(np.int64).sum()) if bits.size else 0
    x = int(np.bitwise_xor.reduce(bits.view(np.uint32))) if bits.size else 0
    return np.array([total, x, bits.size], dtype=np.int64)


def sample_indices(numel: int, n: int = 64, seed: int = 0):
    gen = torch.Generator().manual_seed(seed + numel)
    return torch.randint(0, numel, (min(n, numel),), generator=gen)
