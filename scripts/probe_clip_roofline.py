"""The ClipLoss score contraction alone: bench.py's `roofline_clip` block (kernel / forward microseconds at the cfg2,
cfg3 and 2 048-candidate cfg4 shapes) without the rest of the bench.  BM_BENCH_ZERO_OPERANDS=1 runs the same launches
on all-zero operands (DVFS probe); BM_CLIP_SCORES_KERNEL=0 selects the generic bm_gemm_nt_h2 tiles.

    python scripts/probe_clip_roofline.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import brainmagick_amd  # noqa: E402

brainmagick_amd.set_compute_dtype("f16x2")
for name, v in bench.clip_roofline(torch.device("cuda"), "f16x2", reps=20).items():
    print(f"{name:16s} kernel {v['kernel_us']:8.1f} us  forward {v['forward_us']:8.1f} us  mfma {v['mfma_frac']:.3f} "
          f"hbm {v['hbm_frac']:.3f}  launches {v['launches_per_forward']}")
