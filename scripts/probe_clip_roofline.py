import sys, json, torch
sys.path.insert(0, '.')
import bench, brainmagick_amd
brainmagick_amd.set_compute_dtype("f16x2")
r = bench.clip_roofline(torch.device("cuda"), "f16x2", reps=20)
for k, v in r.items():
    print(f"{k:16s} kernel {v['kernel_us']:8.1f} us  forward {v['forward_us']:8.1f} us  mfma {v['mfma_frac']:.3f} hbm {v['hbm_frac']:.3f}  launches {v['launches_per_forward']}")
