"""Flat-bucket Adam for the hot path.

``torch.optim.Adam(params, lr, betas=(0.9, beta2))`` of bm/train.py:118-119 updates 58 tensors with
one (foreach) kernel chain; here all parameters and all gradients live in ONE flat fp32 buffer each
(the parameters become views into it), so that

  * the optimiser is a single fused HIP launch (``bm_adam_step``), and
  * the data-parallel gradient exchange is one RCCL reduce-scatter + one all-gather on that bucket
    (``distrib.sharded_step``) with each rank updating only its shard (ZeRO-1 style) --
    instead of flashy's per-tensor all-reduces (bm/solver.py:386).

``FlatAdam`` is a ``torch.optim.Optimizer``: ``param_groups`` / ``state`` / ``state_dict`` /
``load_state_dict`` / ``zero_grad`` / ``step`` behave like ``torch.optim.Adam``'s, so the reference's
checkpoint code (bm/solver.py:64,115-117 -- ``self.optimizer.state_dict()`` goes into the flashy
checkpoint) works unchanged and optimizer checkpoints are interchangeable with the reference:
``state[p] = {step, exp_avg, exp_avg_sq}`` where the moments are VIEWS into the flat buffers.
In a data-parallel run every rank only updates the moments of its own shard.  ``state_dict()`` never
communicates (checkpoint code usually calls it on rank 0 only -- a hidden all-gather there would hang or,
worse, pair up with the other ranks' next reduce-scatter): EVERY rank calls ``gather_moments()`` (a
collective; ``Solver.state_dict()`` does) before any rank saves, and ``state_dict()`` raises while the
moments are still sharded.  ``distrib.sharded_step(..., shard=False)`` (or ``BM_SHARD_OPTIMIZER=0``) keeps
the full moments on every rank instead (all-reduce + full Adam), for code that cannot add that call.
"""
import typing as tp

import torch

from . import hip_ops as H


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params: tp.Iterable[torch.nn.Parameter], lr: float = 3e-4,
                 betas: tp.Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, pad_to: int = 1):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("FlatAdam got an empty parameter list")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam runs on the MI355X HIP path only: move the model to the GPU "
                               "before building the optimizer (bm/train.py:89 does)")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdam supports a single parameter group (bm/train.py:118-119)")
        self.params: tp.List[torch.nn.Parameter] = list(self.param_groups[0]["params"])
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
        self._step_tensor = torch.tensor(0.)
        # a data-parallel step leaves the moments outside the own shard stale until gathered
        self._moments_sharded = False
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            self.state[p] = dict(step=self._step_tensor,
                                 exp_avg=self.exp_avg[off:off + k].view(p.shape),
                                 exp_avg_sq=self.exp_avg_sq[off:off + k].view(p.shape))

    # the scalar hyper-parameters live in param_groups like in torch.optim.Adam (lr schedulers work)
    @property
    def lr(self) -> float:
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, value: float):
        self.param_groups[0]["lr"] = value

    @property
    def betas(self) -> tp.Tuple[float, float]:
        return tuple(self.param_groups[0]["betas"])

    @property
    def eps(self) -> float:
        return self.param_groups[0]["eps"]

    # -- torch.optim API used by the Solver (bm/solver.py:384-387) --
    def zero_grad(self, set_to_none: bool = False):
        """``set_to_none=False`` (torch semantics): the bucket is zeroed and every ``p.grad`` is its view of the
        bucket, so autograd accumulates in place (one add per parameter).  ``set_to_none=True``: ``p.grad = None``;
        autograd then hands over its gradient tensors and ``collect_grads()`` moves them into the bucket with one
        multi-tensor copy -- what the Solver does every step."""
        if set_to_none:
            for p, off in zip(self.params, self.offsets):
                p.grad = None
                # the weight-gradient kernels may write straight into the bucket (hip_ops.grad_destination)
                dst = getattr(p, "_bm_grad_dst", None)
                if dst is None or dst[0].data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                    p._bm_grad_dst = (self.flat_grad[off:off + p.numel()].view(p.shape), [False])
                else:
                    dst[1][0] = False
            return
        self.flat_grad.zero_()
        self._attach_views()

    def writing_grads(self):
        """``with optimizer.writing_grads(): loss.backward()`` -- inside, the weight-gradient kernels may write a
        parameter's gradient straight into its slice of the flat bucket (after ``zero_grad(set_to_none=True)``; the
        tensor autograd then stores in ``p.grad`` IS the bucket view, ``collect_grads`` has nothing to copy).  Outside
        such a block (``torch.autograd.grad`` for analysis, a backward pass of some other loss) gradients are fresh
        tensors that alias nothing."""
        return H.direct_grads_armed()

    def _attach_views(self):
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + p.numel()].view(p.shape)

    @torch.no_grad()
    def collect_grads(self):
        """After ``zero_grad(set_to_none=True)`` + backward: copy the gradients autograd produced into the flat
        bucket (parameters that received none read as zero) and make every ``p.grad`` the bucket view again."""
        views, grads = [], []
        for p, off in zip(self.params, self.offsets):
            view = self.flat_grad[off:off + p.numel()].view(p.shape)
            g = p.grad
            if g is None:
                view.zero_()
            elif g.data_ptr() != view.data_ptr():
                views.append(view)
                grads.append(g if g.dtype == torch.float32 else g.float())
            p.grad = view
        if views:
            torch._foreach_copy_(views, grads)

    @torch.no_grad()
    def step(self, closure=None, shard: tp.Optional[tp.Tuple[int, int]] = None,
             grad_scale: float = 1.0):
        """Update the whole bucket, or only elements [lo, hi) when ``shard`` is given."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.collect_grads()                # no-op when every p.grad already is its bucket view
        self.step_count += 1
        self._step_tensor.fill_(float(self.step_count))
        lo, hi = shard if shard is not None else (0, self.padded)
        if shard is not None and (lo, hi) != (0, self.padded):
            self._moments_sharded = True
        if hi > lo:
            b1, b2 = self.betas
            H.adam_step(self.flat_param[lo:hi], self.flat_grad[lo:hi], self.exp_avg[lo:hi],
                        self.exp_avg_sq[lo:hi], self.step_count, self.lr, b1, b2, self.eps, grad_scale)
        return loss

    def gather_moments(self):
        """Data-parallel runs: make ``exp_avg`` / ``exp_avg_sq`` complete on every rank (each rank
        owns the moments of its shard only).  COLLECTIVE: every rank calls it (two in-place all-gathers),
        e.g. at the end of an epoch right before the checkpoint.  No-op for a single process."""
        from . import distrib
        if self._moments_sharded and distrib.is_distributed():
            distrib.all_gather_shards(self.exp_avg)
            distrib.all_gather_shards(self.exp_avg_sq)
        self._moments_sharded = False

    def state_dict(self):
        """torch.optim.Adam layout; moments are cloned (a checkpoint must not alias the live buckets).
        Never communicates: raises while a data-parallel run's moments are still sharded."""
        from . import distrib
        if self._moments_sharded and distrib.is_distributed() and distrib.world_size() > 1:
            raise RuntimeError(
                "FlatAdam.state_dict(): the Adam moments are sharded over the ranks (ZeRO-1); call "
                "optimizer.gather_moments() on EVERY rank first (Solver.state_dict() does), or run "
                "distrib.sharded_step(..., shard=False) / BM_SHARD_OPTIMIZER=0")
        state = {}
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            k = p.numel()
            state[i] = dict(step=torch.tensor(float(self.step_count)),
                            exp_avg=self.exp_avg[off:off + k].view(p.shape).clone(),
                            exp_avg_sq=self.exp_avg_sq[off:off + k].view(p.shape).clone())
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self.params)))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        group = sd["param_groups"][0]
        for key in ("lr", "betas", "eps"):
            self.param_groups[0][key] = tuple(group[key]) if key == "betas" else group[key]
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            st = sd["state"].get(i)
            if st is None:
                continue
            k = p.numel()
            self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = int(st["step"])
        self._step_tensor.fill_(float(self.step_count))
        self._moments_sharded = False      # every rank loaded the complete moments
