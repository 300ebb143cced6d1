"""Data-parallel exchange of the hot path: the replacement for the ``flashy.distrib`` calls on the
training step: one process per GPU, ``torch.distributed`` ("nccl" backend = RCCL
over xGMI on ROCm; "gloo" for the CPU tests).

* C3 ``sync_model`` (bm/solver.py:386): ``sync_flat_gradients`` -- ONE reduce-scatter on the flat
  gradient bucket, the rank updates its shard with the fused Adam, then ONE all-gather of the
  updated parameters.  Same bytes on the wire as an all-reduce, 1/N of the optimizer work, two
  large collectives instead of 58 small ones (xGMI is point-to-point: few, large messages).
* C7 (new vs the reference, which keeps negatives per-GPU -- README.md:139-143): ``CandidateGather``
  all-gathers the precomputed audio candidates of every rank on a side stream, overlapped with the
  SimpleConv forward (candidates are inputs: no dependence on the model), in rank
  order; a rank's targets are its own block, selected with ``ClipLoss.forward(..., target_offset=
  rank*B)`` (the reference contract "first B candidates are the targets", bm/losses.py:105-111, is
  the offset-0 case) -- no re-ordering copy of the (up to 3 GB) gathered tensor.
* C4 ``average_metrics`` (bm/solver.py:395): tiny all-reduce.

Everything degrades to a no-op at world_size 1.
"""
import os
import typing as tp

import torch
import torch.distributed as dist


def _forced() -> bool:
    # BM_FORCE_DISTRIBUTED=1 runs the collective code path even at world_size 1 (used by the GPU
    # test that exercises the real RCCL calls on the single-GPU test box).
    return os.environ.get("BM_FORCE_DISTRIBUTED", "0") == "1"


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _forced())


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def init(backend: tp.Optional[str] = None):
    """flashy.distrib.init (bm/train.py:139): rendezvous from the torchrun environment."""
    if "RANK" not in os.environ or (int(os.environ.get("WORLD_SIZE", "1")) == 1 and not _forced()):
        return
    if dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=backend)


def barrier():
    if is_distributed():
        dist.barrier()


def shard_bounds(numel: int, world: int, r: int) -> tp.Tuple[int, int]:
    """Equal shards of a bucket whose length is a multiple of ``world``."""
    assert numel % world == 0, "pad the flat bucket to a multiple of the world size"
    per = numel // world
    return r * per, (r + 1) * per


def _reduce_scatter_sum(flat: torch.Tensor, world: int, r: int) -> None:
    """In place: afterwards shard r of ``flat`` holds the sum over ranks of that shard."""
    lo, hi = shard_bounds(flat.numel(), world, r)
    if dist.get_backend() == "gloo":
        # gloo has no reduce_scatter: all-reduce then keep the own shard (CPU tests only)
        dist.all_reduce(flat)
        return
    out = torch.empty(hi - lo, device=flat.device, dtype=flat.dtype)
    dist.reduce_scatter_tensor(out, flat)
    flat[lo:hi].copy_(out)


def sync_flat_gradients(optimizer, average: bool = True) -> tp.Optional[tp.Tuple[int, int]]:
    """Reduce-scatter the flat gradient bucket; returns this rank's shard bounds (None when not
    distributed).  Gradient averaging (flashy sync_model semantics) is folded in the Adam kernel
    through ``grad_scale`` by ``sharded_step``."""
    if not is_distributed():
        return None
    world, r = world_size(), rank()
    _reduce_scatter_sum(optimizer.flat_grad, world, r)
    return shard_bounds(optimizer.flat_grad.numel(), world, r)


def sharded_step(optimizer) -> None:
    """``flashy.distrib.sync_model`` + ``optimizer.step()`` (bm/solver.py:386-387) on the flat bucket:
    reduce-scatter(grads) -> Adam on the own shard (mean over ranks via grad_scale) -> all-gather
    (params)."""
    if not is_distributed():
        optimizer.step()
        return
    world = world_size()
    shard = sync_flat_gradients(optimizer)
    optimizer.step(shard=shard, grad_scale=1.0 / world)
    lo, hi = shard
    dist.all_gather_into_tensor(optimizer.flat_param, optimizer.flat_param[lo:hi].clone())


def sync_buffers(model: torch.nn.Module) -> None:
    """flashy.distrib.sync_model also averages float buffers (BatchNorm running statistics)
    [upstream, unverified]; one flat all-reduce."""
    if not is_distributed():
        return
    bufs = [b for b in model.buffers() if b.is_floating_point()]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1) for b in bufs])
    dist.all_reduce(flat)
    flat /= world_size()
    off = 0
    for b in bufs:
        b.copy_(flat[off:off + b.numel()].view_as(b))
        off += b.numel()


def average_metrics(metrics: tp.Dict[str, float], count: float = 1.) -> tp.Dict[str, float]:
    """flashy.distrib.average_metrics (bm/solver.py:395): weighted mean over ranks."""
    if not is_distributed():
        return dict(metrics)
    keys = sorted(metrics)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(metrics[k]) * count for k in keys] + [float(count)], device=dev,
                     dtype=torch.float64 if dev == "cpu" else torch.float32)
    dist.all_reduce(t)
    return {k: (t[i] / t[-1]).item() for i, k in enumerate(keys)}


class CandidateGather:
    """Whole-node negatives: all-gather of the candidate features, overlapped with the forward.

    ``start(candidates)`` enqueues the all-gather on a side stream; ``wait()`` returns
    ``(gathered [world*B, ...] in rank order, target_offset = rank*B)``."""

    def __init__(self):
        self.stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        self._out = None
        self._work = None
        self._B = 0

    def start(self, candidates: torch.Tensor):
        if not is_distributed():
            self._out = candidates
            return
        world = world_size()
        self._B = candidates.shape[0]
        candidates = candidates.contiguous()
        out = torch.empty((world * self._B,) + tuple(candidates.shape[1:]), device=candidates.device,
                          dtype=candidates.dtype)
        if self.stream is not None and candidates.is_cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                dist.all_gather_into_tensor(out, candidates)
            candidates.record_stream(self.stream)
        else:
            dist.all_gather_into_tensor(out, candidates)
        self._out = out

    def wait(self) -> tp.Tuple[torch.Tensor, int]:
        out = self._out
        self._out = None
        if not is_distributed():
            return out, 0
        if self.stream is not None and out.is_cuda:
            torch.cuda.current_stream().wait_stream(self.stream)
        return out, rank() * self._B


class GatherCandidatesFn(torch.autograd.Function):
    """All-gather of LEARNABLE candidates (feature model on, bm/solver.py:304-320) with its adjoint:
    the gradient of the gathered tensor is reduce-scattered (summed over ranks) back to the owner of
    each block.  Together with the mean-over-ranks of parameter gradients this yields the gradient
    of the mean of the per-rank losses."""

    @staticmethod
    def forward(ctx, candidates):
        world = world_size()
        candidates = candidates.contiguous()
        out = torch.empty((world * candidates.shape[0],) + tuple(candidates.shape[1:]),
                          device=candidates.device, dtype=candidates.dtype)
        dist.all_gather_into_tensor(out, candidates)
        return out

    @staticmethod
    def backward(ctx, grad):
        world, r = world_size(), rank()
        grad = grad.contiguous()
        B = grad.shape[0] // world
        if dist.get_backend() == "gloo":
            dist.all_reduce(grad)
            return grad[r * B:(r + 1) * B].clone()
        out = torch.empty((B,) + tuple(grad.shape[1:]), device=grad.device, dtype=grad.dtype)
        dist.reduce_scatter_tensor(out, grad)
        return out


def gather_learnable_candidates(candidates: torch.Tensor):
    """-> (gathered [world*B, ...], target_offset); identity at world_size 1."""
    if not is_distributed():
        return candidates, 0
    return GatherCandidatesFn.apply(candidates), rank() * candidates.shape[0]
