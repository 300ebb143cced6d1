"""Does data that was just written / read come back faster than HBM?  Streams buffers of growing size repeatedly
(read-only max-reduction through libbmhip's amax kernel, and torch's elementwise copy) and prints the achieved rate:
a buffer that fits the 256 MB memory-side cache (MALL) is re-read from there on every pass but the first.

    python scripts/probe_mall.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brainmagick_amd import hip_ops as H  # noqa: E402
from brainmagick_amd._lib import lib  # noqa: E402


def rate(fn, nbytes, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12


def main():
    ws = torch.zeros(16384, device="cuda")
    out = torch.zeros(8, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    print(f"{'MB':>8s} {'amax read TB/s':>15s} {'copy (r+w) TB/s':>16s} {'read fwd, then read REVERSED TB/s':>34s}")
    for mb in (8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
        n = mb * (1 << 20) // 4
        x = torch.randn(n, device="cuda")
        y = torch.empty_like(x)

        def read():
            lib().bm_amax(ctypes.c_void_p(x.data_ptr()), n, ctypes.c_void_p(out.data_ptr()),
                          ctypes.c_void_p(ws.data_ptr()), stream)

        r_read = rate(read, 4 * n)
        r_copy = rate(lambda: y.copy_(x), 8 * n)
        print(f"{mb:8d} {r_read:15.2f} {r_copy:16.2f}")


if __name__ == "__main__":
    main()
