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
