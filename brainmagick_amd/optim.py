"""Flat-bucket Adam for the hot path.

``torch.optim.Adam(params, lr, betas=(0.9, beta2))`` of bm/train.py:118-119 updates 58 tensors with
one (foreach) kernel chain; here all parameters and all gradients live in ONE flat fp32 buffer each
(the parameters become views into it), so that

  * the optimiser is a single fused HIP launch (``bm_adam_step``), and
  * the data-parallel gradient exchange is one RCCL reduce-scatter + one all-gather on that bucket
    (``distrib.sync_flat_gradients``) with each rank updating only its shard (ZeRO-1 style) --
    instead of flashy's per-tensor all-reduces (bm/solver.py:386).

``state_dict`` / ``load_state_dict`` use torch.optim.Adam's layout (per-parameter ``step``,
``exp_avg``, ``exp_avg_sq``) so optimizer checkpoints stay interchangeable with the reference.
"""
import typing as tp

import torch

from . import hip_ops as H


class FlatAdam:
    def __init__(self, params: tp.Iterable[torch.nn.Parameter], lr: float = 3e-4,
                 betas: tp.Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, pad_to: int = 1):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdam got an empty parameter list")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam runs on the MI355X HIP path only: move the model to the GPU "
                               "before building the optimizer (bm/train.py:89 does)")
        self.lr, self.betas, self.eps = lr, betas, eps
        n = sum(p.numel() for p in self.params)
        self.numel = n
        self.padded = (n + pad_to - 1) // pad_to * pad_to
        self.flat_param = torch.zeros(self.padded, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(self.padded, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(self.padded, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(self.padded, device=dev, dtype=torch.float32)
        self.offsets: tp.List[int] = []
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat_param[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + k].view(p.shape)
            p.grad = self.flat_grad[off:off + k].view(p.shape)
            self.offsets.append(off)
            off += k
        self.step_count = 0

    # -- torch.optim API subset used by the Solver (bm/solver.py:384-387) --
    def zero_grad(self, set_to_none: bool = False):
        self.flat_grad.zero_()
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + p.numel()].view(p.shape)

    def step(self, shard: tp.Optional[tp.Tuple[int, int]] = None, grad_scale: float = 1.0):
        """Update the whole bucket, or only elements [lo, hi) when ``shard`` is given."""
        self.step_count += 1
        lo, hi = shard if shard is not None else (0, self.padded)
        if hi > lo:
            H.adam_step(self.flat_param[lo:hi], self.flat_grad[lo:hi], self.exp_avg[lo:hi],
                        self.exp_avg_sq[lo:hi], self.step_count, self.lr, self.betas[0],
                        self.betas[1], self.eps, grad_scale)

    def state_dict(self):
        state = {}
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            k = p.numel()
            state[i] = dict(step=torch.tensor(float(self.step_count)),
                            exp_avg=self.exp_avg[off:off + k].view(p.shape).clone(),
                            exp_avg_sq=self.exp_avg_sq[off:off + k].view(p.shape).clone())
        group = dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=0, amsgrad=False,
                     params=list(range(len(self.params))))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        group = sd["param_groups"][0]
        self.lr, self.betas, self.eps = group["lr"], tuple(group["betas"]), group["eps"]
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            st = sd["state"].get(i)
            if st is None:
                continue
            k = p.numel()
            self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = int(st["step"])
