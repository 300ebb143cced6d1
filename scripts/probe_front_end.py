"""Wall time (with a synchronisation around every call) of each library call of the composed front end
(functional.FusedFrontEndFn, forward + backward) at the cfg2 shapes: which of its small launches matter.

    python scripts/probe_front_end.py
"""
import sys, time, torch
sys.path.insert(0, '.')
import brainmagick_amd
from brainmagick_amd import hip_ops as H, functional as BF
brainmagick_amd.set_compute_dtype("f16x2")
torch.manual_seed(0)
B, C, T, O, Dp, L, S, D, U = 256, 208, 360, 270, 288, 270, 27, 270, 1
dev = "cuda"
meg = torch.randn(B, C, T, device=dev)
heads = torch.randn(O, Dp, device=dev, requires_grad=True)
w1 = (torch.randn(L, O, 1, device=dev) / 16).requires_grad_()
b1 = torch.randn(L, device=dev, requires_grad=True)
ws = (torch.randn(S, L, D, device=dev) / 16).requires_grad_()
pos = torch.rand(U, C, 2, device=dev)
lay = torch.zeros(B, dtype=torch.int64, device=dev)
sub = torch.randint(0, S, (B,), device=dev)
ban = torch.tensor([0.5, 0.5], device=dev)
# wrap the library calls with per-call timing
log = []
def wrap(name):
    fn = getattr(H, name)
    def inner(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); log.append((name, [tuple(x.shape) if torch.is_tensor(x) else x for x in a[:2]] + list(a[2:8]), (time.perf_counter() - t0) * 1e6))
        return r
    setattr(H, name, inner)
for n in ("gemm_nt", "conv_nn", "pack_weights", "sum_over_batch", "segment_sum_cols", "time_sums_t", "softmax_bwd", "masked_softmax", "group_by_index", "fourier_emb"):
    wrap(n)
for it in range(3):
    log.clear()
    out = BF.FusedFrontEndFn.apply(meg, heads, w1, b1, ws, pos, lay, sub, ban, 0.0)
    nf = len(log)
    out.backward(torch.randn_like(out))
for i, (n, a, us) in enumerate(log):
    print("FWD" if i < nf else "BWD", f"{n:18s} {us:8.1f} us  {str(a)[:150]}")
