"""Data-parallel exchange of the hot path: the replacement for the ``flashy.distrib`` calls on the
training step, one process per GPU.

Data plane on the GPU = RCCL behind the C-ABI (``bm_comm_*`` in include/bm_hip.h, csrc/comm.hip):
the 128-byte unique id travels through the rendezvous TCP store (MASTER_ADDR / MASTER_PORT of the
torchrun environment), every collective is enqueued on a HIP stream by libbmhip -- PyTorch only owns
the buffers.  ``torch.distributed`` is used for the CPU tests ("gloo") and as an explicit fallback
(``BM_COMM=torch`` or ``init("nccl")``).

* C3 ``sync_model`` (bm/solver.py:386): ``sharded_step`` -- ONE in-place reduce-scatter on the flat
  gradient bucket, the rank updates its shard with the fused Adam (mean over ranks folded in as
  ``grad_scale``), then ONE in-place all-gather of the updated parameters.  Same bytes on the wire
  as an all-reduce, 1/N of the optimizer work, two large collectives instead of 58 small ones
  (xGMI is point-to-point: few, large messages).  Float buffers (BatchNorm running statistics) are
  averaged like flashy's ``sync_model`` does, through ONE flat all-reduce (``BufferBucket``).
* C7 (new vs the reference, which keeps negatives per-GPU -- README.md:139-143): ``CandidateGather``
  all-gathers the precomputed audio candidates of every rank on a side stream, overlapped with the
  SimpleConv forward (candidates are inputs: no dependence on the model), in rank order; a rank's
  targets are its own block, selected with ``ClipLoss.forward(..., target_offset=rank*B)`` (the
  reference contract "first B candidates are the targets", bm/losses.py:105-111, is the offset-0
  case) -- no re-ordering copy of the (up to 3 GB) gathered tensor.
* C4 ``average_metrics`` (bm/solver.py:395): tiny all-reduce.

Everything degrades to a no-op at world_size 1.
"""
import contextlib
import ctypes
import datetime
import os
import sys
import typing as tp

import torch
import torch.distributed as dist


def _forced() -> bool:
    # BM_FORCE_DISTRIBUTED=1 runs the collective code path even at world_size 1 (used by the GPU
    # test that exercises the real RCCL calls on the single-GPU test box).
    return os.environ.get("BM_FORCE_DISTRIBUTED", "0") == "1"


# ------------------------------------------------------------------------------------------------
# communicators
# ------------------------------------------------------------------------------------------------
class _TorchComm:
    """torch.distributed process group: "gloo" (CPU tests) or "nccl" (= RCCL through PyTorch, fallback)."""

    def __init__(self, backend: str, store=None):
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        if backend == "gloo" and os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # the container hostname may not resolve
        if not dist.is_initialized():
            if store is not None:
                # the rendezvous store this process already holds (outside torchrun rank 0 HOSTS it on
                # MASTER_PORT: a second env:// rendezvous would die with EADDRINUSE)
                dist.init_process_group(backend=backend, store=dist.PrefixStore("bm_pg", store),
                                        rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
            else:
                dist.init_process_group(backend=backend)
        self.backend = backend
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.kind = f"torch.distributed/{backend}"

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        dist.all_gather_into_tensor(out, inp)

    def reduce_scatter_shard(self, flat: torch.Tensor):
        """In place: afterwards shard `rank` of ``flat`` holds the sum over ranks of that shard."""
        lo, hi = shard_bounds(flat.numel(), self.world, self.rank)
        if self.backend == "gloo":
            dist.all_reduce(flat)          # gloo has no reduce_scatter (CPU tests only)
            return
        out = torch.empty(hi - lo, device=flat.device, dtype=flat.dtype)
        dist.reduce_scatter_tensor(out, flat)
        flat[lo:hi].copy_(out)

    def all_gather_shards(self, flat: torch.Tensor):
        lo, hi = shard_bounds(flat.numel(), self.world, self.rank)
        dist.all_gather_into_tensor(flat, flat[lo:hi].clone())

    def reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor):
        if self.backend == "gloo":
            tmp = inp.clone()
            dist.all_reduce(tmp)
            n = out.numel()
            out.copy_(tmp.reshape(-1)[self.rank * n:(self.rank + 1) * n].view_as(out))
            return
        dist.reduce_scatter_tensor(out, inp)

    def all_reduce(self, t: torch.Tensor, op: str = "sum"):
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)

    def barrier(self):
        dist.barrier()

    def scalar_device(self):
        return "cuda" if self.backend == "nccl" else "cpu"

    def reported_world(self) -> int:
        return dist.get_world_size()

    def close(self):
        if dist.is_initialized():
            dist.destroy_process_group()


_store_cache: tp.Optional[tp.Any] = None


def _rendezvous_store(rank: int, world: int):
    """The TCP store of the torchrun / env:// rendezvous (the agent's store under torchrun, else rank 0
    hosts it on MASTER_ADDR:MASTER_PORT).  Host-side key/value plumbing only; one per process."""
    global _store_cache
    if _store_cache is None and dist.is_available() and dist.is_initialized():
        # somebody already ran the env:// rendezvous in this process (outside torchrun rank 0 hosts the store on
        # MASTER_PORT: a second TCPStore there would die with EADDRINUSE): ride on that group's store
        try:
            from torch.distributed.distributed_c10d import _get_default_store
            _store_cache = dist.PrefixStore("bm_rdzv", _get_default_store())
        except Exception:                   # no accessible default store: fall through to our own
            _store_cache = None
    if _store_cache is None:
        host = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("MASTER_PORT", "29500"))
        agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
        _store_cache = dist.TCPStore(host, port, world, is_master=(rank == 0 and not agent),
                                     timeout=datetime.timedelta(seconds=300), wait_for_workers=False)
    return _store_cache


def _all_ranks_ok(store, key: str, rank: int, world: int, ok: bool) -> bool:
    """Every rank publishes whether a LOCAL step worked and reads everybody's answer: the communicator choice
    (RCCL behind the C-ABI or the torch.distributed fallback) must be the same on every rank, or the job
    deadlocks with mixed communicators."""
    store.set(f"{key}/{rank}", b"1" if ok else b"0")
    return all(bytes(store.get(f"{key}/{r}")) == b"1" for r in range(world))


@contextlib.contextmanager
def _c_stdout_to_stderr():
    """librccl prints a version banner to the C-level stdout when the first communicator comes up.  A tool that
    prints a machine-readable line on stdout (bench.py) must not have library chatter in front of it: while the
    communicator is created, file descriptor 1 points at stderr."""
    try:
        sys.stdout.flush()
        libc = ctypes.CDLL(None)
        libc.fflush(None)
        saved = os.dup(1)
    except Exception:                       # no usable fd 1 (embedded interpreter): nothing to protect
        yield
        return
    try:
        os.dup2(2, 1)
        yield
    finally:
        try:
            libc.fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(saved)


class RcclUnavailable(RuntimeError):
    """RCCL through the C-ABI cannot be used on at least one rank (agreed through the rendezvous store)."""


class _RcclComm:
    """RCCL behind the C-ABI (bm_comm_*)."""
    _generation = 0

    def __init__(self):
        from ._lib import lib, check
        self._lib, self._check = lib(), check
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.device = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(self.device)
        rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if "BM_RCCL_LIB" not in os.environ and os.path.exists(rccl):
            os.environ["BM_RCCL_LIB"] = rccl           # the copy PyTorch ships and tests on this driver
        nbytes = self._lib.bm_comm_unique_id_bytes()
        gen = _RcclComm._generation
        _RcclComm._generation += 1
        key = f"bm_comm/{gen}"
        self._h = None
        self._store = _rendezvous_store(self.rank, self.world) if self.world > 1 else None
        # phase 1 (local, cannot hang): librccl loads and exports what comm.hip binds -- agreed over all ranks
        # BEFORE anybody enters the collective ncclCommInitRank
        loaded = self._lib.bm_comm_available() == 0
        if self._store is not None:
            loaded = _all_ranks_ok(self._store, key + "/loaded", self.rank, self.world, loaded)
        if not loaded:
            raise RcclUnavailable("librccl could not be loaded on at least one rank: " +
                                  (self._lib.bm_last_error() or b"").decode())
        # the unique id: rank 0 creates and publishes it.  A failure there is published too (an empty value), so that
        # the other ranks fall back with rank 0 instead of sitting in store.get() until the store times out
        if self.rank == 0:
            uid, err = b"", ""
            try:
                buf = ctypes.create_string_buffer(nbytes)
                check(self._lib.bm_comm_unique_id(buf), "bm_comm_unique_id")
                uid = buf.raw
            except Exception as exc:            # noqa: BLE001 -- reported to every rank through the store
                err = str(exc)
            if self._store is not None:
                self._store.set(key + "/id", uid)
            if not uid:
                raise RcclUnavailable(f"bm_comm_unique_id failed on rank 0: {err}")
        else:
            uid = bytes(self._store.get(key + "/id"))
            if not uid:
                raise RcclUnavailable("bm_comm_unique_id failed on rank 0")
        # phase 2: the communicator itself; the outcome is agreed again so that a rank whose init failed does
        # not leave the others with a communicator nobody else joins
        handle = ctypes.c_void_p()
        with _c_stdout_to_stderr():
            rc = self._lib.bm_comm_init(uid, self.world, self.rank, self.device, ctypes.byref(handle))
        ok = rc == 0
        msg = "" if ok else (self._lib.bm_last_error() or b"").decode()
        if self._store is not None:
            ok = _all_ranks_ok(self._store, key + "/init", self.rank, self.world, ok)
        if not ok:
            if rc == 0:
                self._lib.bm_comm_destroy(handle)
            raise RcclUnavailable(f"ncclCommInitRank failed on at least one rank {msg}")
        self._h = handle
        self.kind = "rccl/c-abi"

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError(f"bm_comm.{name}: expected a contiguous fp32 GPU tensor")
        return t

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        self._f32(out, "allgather"), self._f32(inp, "allgather")
        assert out.numel() == self.world * inp.numel()
        self._check(self._lib.bm_comm_allgather(self._h, ctypes.c_void_p(inp.data_ptr()),
                                                ctypes.c_void_p(out.data_ptr()), inp.numel(), self._stream()),
                    "bm_comm_allgather")

    def reduce_scatter_shard(self, flat: torch.Tensor):
        self._f32(flat, "reduce_scatter")
        lo, hi = shard_bounds(flat.numel(), self.world, self.rank)
        # in place: recv == send + rank * count
        self._check(self._lib.bm_comm_reduce_scatter(self._h, ctypes.c_void_p(flat.data_ptr()),
                                                     ctypes.c_void_p(flat.data_ptr() + 4 * lo), hi - lo,
                                                     self._stream()), "bm_comm_reduce_scatter")

    def all_gather_shards(self, flat: torch.Tensor):
        self._f32(flat, "allgather")
        lo, hi = shard_bounds(flat.numel(), self.world, self.rank)
        # in place: send == recv + rank * count
        self._check(self._lib.bm_comm_allgather(self._h, ctypes.c_void_p(flat.data_ptr() + 4 * lo),
                                                ctypes.c_void_p(flat.data_ptr()), hi - lo, self._stream()),
                    "bm_comm_allgather")

    def reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor):
        self._f32(out, "reduce_scatter"), self._f32(inp, "reduce_scatter")
        assert inp.numel() == self.world * out.numel()
        self._check(self._lib.bm_comm_reduce_scatter(self._h, ctypes.c_void_p(inp.data_ptr()),
                                                     ctypes.c_void_p(out.data_ptr()), out.numel(),
                                                     self._stream()), "bm_comm_reduce_scatter")

    def all_reduce(self, t: torch.Tensor, op: str = "sum"):
        self._f32(t, "allreduce")
        self._check(self._lib.bm_comm_allreduce(self._h, ctypes.c_void_p(t.data_ptr()),
                                                ctypes.c_void_p(t.data_ptr()), t.numel(),
                                                0 if op == "sum" else 1, self._stream()), "bm_comm_allreduce")

    def barrier(self):
        t = torch.zeros(1, device="cuda", dtype=torch.float32)
        self.all_reduce(t)
        torch.cuda.current_stream().synchronize()

    def scalar_device(self):
        return "cuda"

    def reported_world(self) -> int:
        """ncclCommCount of the live communicator: the world size as RCCL itself reports it."""
        return int(self._lib.bm_comm_reported_world(self._h))

    def close(self):
        if self._h:
            torch.cuda.synchronize()
            self._lib.bm_comm_destroy(self._h)
            self._h = None


_comm: tp.Optional[tp.Any] = None


def comm():
    return _comm


class CommTimer:
    """Optional per-phase timing of the step's collectives with HIP events ON THE STREAM EACH ONE IS ENQUEUED ON
    (``bench.py --gpus N`` installs one for its event pass, so that a first real multi-GPU run explains itself):

      reduce_scatter   gradients, in place on the flat bucket          (compute stream)
      adam_shard       fused Adam on the own shard                     (compute stream; not communication, for scale)
      all_gather       updated parameters, in place                    (compute stream)
      buffer_allreduce BatchNorm running statistics                    (compute stream)
      cand_gather      candidate all-gather of the whole-node negatives (side stream, overlapped with the backward pass)
      gather_wait      what the compute stream actually WAITS for that gather when ClipLoss needs it (the exposed part)
    """

    def __init__(self):
        self.records: tp.List[tp.Tuple[str, torch.cuda.Event, torch.cuda.Event]] = []

    @contextlib.contextmanager
    def phase(self, name: str):
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()                      # on the CURRENT stream of the caller's context
        try:
            yield
        finally:
            end.record()
            self.records.append((name, start, end))

    def summary(self, steps: int) -> tp.Dict[str, tp.Dict[str, float]]:
        """phase -> {calls, ms_per_step, avg_ms}; call after a device synchronize."""
        acc: tp.Dict[str, tp.List[float]] = {}
        for name, start, end in self.records:
            acc.setdefault(name, []).append(start.elapsed_time(end))
        return {k: dict(calls=len(v), ms_per_step=sum(v) / max(steps, 1), avg_ms=sum(v) / len(v))
                for k, v in acc.items()}


_comm_timer: tp.Optional[CommTimer] = None


def set_comm_timer(timer: tp.Optional[CommTimer]) -> None:
    global _comm_timer
    _comm_timer = timer


def _phase(name: str):
    return _comm_timer.phase(name) if (_comm_timer is not None and torch.cuda.is_available()) \
        else contextlib.nullcontext()


def reported_world() -> int:
    """The communicator's size as the communication library reports it (ncclCommCount / torch.distributed)."""
    c = _adopt_torch_group()
    if c is None:
        return 1
    fn = getattr(c, "reported_world", None)
    return int(fn()) if fn is not None else int(c.world)


def comm_kind() -> str:
    return _comm.kind if _comm is not None else "none"


def _adopt_torch_group():
    """A process group somebody else initialised (``flashy.distrib.init()`` at bm/train.py:139 does exactly
    ``torch.distributed.init_process_group``): without this, a maintainer who keeps that call and only swaps in
    ``sharded_step`` would train N replicas with NO gradient exchange, silently.  The group is adopted as the
    communicator the first time anything here asks."""
    global _comm
    if _comm is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        _comm = _TorchComm(str(dist.get_backend()))
    return _comm


def is_distributed() -> bool:
    c = _adopt_torch_group()
    return c is not None and (c.world > 1 or _forced())


def rank() -> int:
    c = _adopt_torch_group()
    return c.rank if c is not None else 0


def world_size() -> int:
    c = _adopt_torch_group()
    return c.world if c is not None else 1


def init(backend: tp.Optional[str] = None):
    """flashy.distrib.init (bm/train.py:139): rendezvous from the torchrun environment
    (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT).

    backend: "rccl" (default on a GPU: RCCL behind the C-ABI), "gloo" (default on CPU), "nccl"
    (torch.distributed's RCCL binding; also chosen with BM_COMM=torch)."""
    global _comm
    if "RANK" not in os.environ or (int(os.environ.get("WORLD_SIZE", "1")) == 1 and not _forced()):
        return
    if _comm is not None:
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "rccl" if torch.cuda.is_available() else "gloo"
        if backend == "rccl" and os.environ.get("BM_COMM", "") == "torch":
            backend = "nccl"
    if dist.is_initialized() and backend in ("gloo", "nccl"):
        _comm = _TorchComm(backend)           # adopt the group that already exists
    elif backend == "rccl":
        try:
            _comm = _RcclComm()
        except RcclUnavailable as exc:        # agreed by every rank through the store: all fall back together
            import warnings
            warnings.warn(f"RCCL through the C-ABI is unavailable ({exc}); using torch.distributed's binding")
            _comm = _TorchComm("nccl", store=_store_cache)
    else:
        _comm = _TorchComm(backend)


def shutdown():
    """Tear the communicator down; a later ``init()`` (another MASTER_PORT / WORLD_SIZE in the same process) starts
    from a fresh rendezvous store."""
    global _comm, _store_cache
    if _comm is not None:
        _comm.close()
        _comm = None
    _store_cache = None


def barrier():
    if is_distributed():
        _comm.barrier()


def shard_bounds(numel: int, world: int, r: int) -> tp.Tuple[int, int]:
    """Equal shards of a bucket whose length is a multiple of ``world``."""
    assert numel % world == 0, "pad the flat bucket to a multiple of the world size"
    per = numel // world
    return r * per, (r + 1) * per


def all_gather_shards(flat: torch.Tensor) -> None:
    """Every rank holds a valid shard `rank` of ``flat``; afterwards all of ``flat`` is valid everywhere."""
    if is_distributed():
        _comm.all_gather_shards(flat)


def sync_flat_gradients(optimizer, average: bool = True) -> tp.Optional[tp.Tuple[int, int]]:
    """Reduce-scatter the flat gradient bucket in place; returns this rank's shard bounds (None when
    not distributed).  Gradient averaging (flashy sync_model semantics) is folded in the Adam kernel
    through ``grad_scale`` by ``sharded_step``."""
    if not is_distributed():
        return None
    _comm.reduce_scatter_shard(optimizer.flat_grad)
    return shard_bounds(optimizer.flat_grad.numel(), world_size(), rank())


def hip_ops_weights_changed():
    from . import hip_ops
    hip_ops.weights_changed()


_SHARD_DEFAULT = os.environ.get("BM_SHARD_OPTIMIZER", "1") == "1"


def sharded_step(optimizer, buffers: tp.Optional["BufferBucket"] = None, shard: tp.Optional[bool] = None) -> None:
    """``flashy.distrib.sync_model`` + ``optimizer.step()`` (bm/solver.py:386-387) on the flat bucket:
    reduce-scatter(grads) -> Adam on the own shard (mean over ranks via grad_scale) -> all-gather
    (params); float buffers (BatchNorm running statistics) averaged with one all-reduce.

    ``shard=False`` (default from ``BM_SHARD_OPTIMIZER=0``): one in-place all-reduce of the gradients and the
    full Adam on every rank -- same bytes on the wire, N times the (tiny) optimizer work, and the Adam moments
    stay complete on every rank, so a rank-0-only ``optimizer.state_dict()`` needs no ``gather_moments()``."""
    if not is_distributed():
        optimizer.step()
        return
    # after zero_grad(set_to_none=True) the bucket is only complete once the gradients autograd handed over have
    # been moved into it: never communicate a partly fresh bucket (no-op when the caller did it already)
    collect = getattr(optimizer, "collect_grads", None)
    if collect is not None:
        collect()
    if shard is None:
        shard = _SHARD_DEFAULT
    if shard:
        with _phase("reduce_scatter"):
            bounds = sync_flat_gradients(optimizer)
        with _phase("adam_shard"):
            optimizer.step(shard=bounds, grad_scale=1.0 / world_size())
        with _phase("all_gather"):
            _comm.all_gather_shards(optimizer.flat_param)
    else:
        with _phase("all_reduce"):
            _comm.all_reduce(optimizer.flat_grad)
        with _phase("adam_full"):
            optimizer.step(grad_scale=1.0 / world_size())
    hip_ops_weights_changed()               # parameters were written through raw pointers: packed copies are stale
    if buffers is not None:
        with _phase("buffer_allreduce"):
            buffers.average()


class BufferBucket:
    """All floating-point buffers of the models (BatchNorm running_mean / running_var) as views into ONE
    flat tensor, so that flashy.distrib.sync_model's buffer averaging [upstream, unverified] is a
    single in-place all-reduce + one scale instead of a cat / copy-back per step."""

    def __init__(self, models: tp.Iterable[torch.nn.Module]):
        self.buffers = [b for m in models for b in m.buffers() if b.is_floating_point()]
        self.flat = None
        if self.buffers:
            dev = self.buffers[0].device
            n = sum(b.numel() for b in self.buffers)
            self.flat = torch.empty(n, device=dev, dtype=torch.float32)
            off = 0
            for b in self.buffers:
                k = b.numel()
                self.flat[off:off + k].copy_(b.reshape(-1))
                b.data = self.flat[off:off + k].view(b.shape)       # same tensor object, new storage
                off += k

    def average(self):
        if self.flat is None or not is_distributed():
            return
        _comm.all_reduce(self.flat)
        self.flat.mul_(1.0 / world_size())


def sync_buffers(model: torch.nn.Module) -> None:
    """One-off form of ``BufferBucket.average`` for a model whose buffers were not flattened."""
    if not is_distributed():
        return
    bufs = [b for b in model.buffers() if b.is_floating_point()]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1) for b in bufs]).float().contiguous()
    _comm.all_reduce(flat)
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
    t = torch.tensor([float(metrics[k]) * count for k in keys] + [float(count)],
                     device=_comm.scalar_device(), dtype=torch.float32)
    _comm.all_reduce(t)
    t = t.double().cpu()
    return {k: (t[i] / t[-1]).item() for i, k in enumerate(keys)}


def max_over_ranks(value: float) -> float:
    """Used by bench.py: the step time of the slowest rank."""
    if not is_distributed():
        return value
    t = torch.tensor([value], device=_comm.scalar_device(), dtype=torch.float32)
    _comm.all_reduce(t, op="max")
    return float(t.cpu())


def check_equal_over_ranks(value: int, what: str) -> None:
    """Raises when ``value`` differs between ranks (two tiny all-reduces: max(v) and max(-v))."""
    if not is_distributed():
        return
    t = torch.tensor([float(value), -float(value)], device=_comm.scalar_device(), dtype=torch.float32)
    _comm.all_reduce(t, op="max")
    hi, lo = float(t[0].cpu()), -float(t[1].cpu())
    if hi != lo:
        raise RuntimeError(f"{what} differs between ranks ({lo:.0f} .. {hi:.0f}): whole-node negatives need the "
                           "same number of segments on every rank")


class CandidateGather:
    """Whole-node negatives: all-gather of the candidate features, overlapped with the forward.

    ``start(candidates)`` enqueues the all-gather on a side stream; ``wait()`` returns
    ``(gathered [world*B, ...] in rank order, target_offset = rank*B, valid | None)``.

    Ranks normally bring the same number of segments (``BM_CHECK_RANKS=1`` verifies it with an extra tiny
    all-reduce).  When per-rank rejection can break that (``ScaleReject`` without clipping, bm/norm.py:335-343), the
    Solver passes ``block_rows`` = the nominal per-rank batch size: a rank's ``n <= block_rows`` candidates are
    zero-padded to the block, the counts travel with a second (one float per rank) all-gather on the same stream,
    and ``valid`` ([world*block_rows] fp32, 1 = real candidate) masks the padding rows in ClipLoss -- all on the
    device, no host synchronisation."""

    def __init__(self):
        self.stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        self._out = None
        self._B = 0
        self._valid = None

    def start(self, candidates: torch.Tensor, block_rows: tp.Optional[int] = None, verify_block: bool = False):
        """``verify_block``: the caller derived ``block_rows`` from its LOCAL batch (not from the configuration), so
        ranks may disagree (the short last batch of an epoch, an unbalanced sampler): agree on it with a tiny all-reduce
        before the sized collective is issued -- an all-gather with unequal blocks hangs or corrupts."""
        self._valid = None
        if not is_distributed():
            self._out = candidates
            return
        world = world_size()
        n = candidates.shape[0]
        self._B = n if block_rows is None else block_rows
        assert n <= self._B, (n, self._B)
        if verify_block or os.environ.get("BM_CHECK_RANKS", "0") == "1":
            check_equal_over_ranks(self._B * (candidates.numel() // max(n, 1)), "candidate block size")
        candidates = candidates.contiguous()
        side = self.stream if (self.stream is not None and candidates.is_cuda) else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            out = torch.empty((world * self._B,) + tuple(candidates.shape[1:]), device=candidates.device,
                              dtype=candidates.dtype)
            send = candidates
            if block_rows is not None:
                if n < self._B:
                    send = candidates.new_zeros((self._B,) + tuple(candidates.shape[1:]))
                    send[:n].copy_(candidates)
                counts = torch.empty(world, device=candidates.device, dtype=torch.float32)
                _comm.all_gather(counts, torch.full((1,), float(n), device=candidates.device, dtype=torch.float32))
                rows = torch.arange(self._B, device=candidates.device, dtype=torch.float32)
                self._valid = (rows[None, :] < counts[:, None]).to(torch.float32).reshape(-1).contiguous()
            with _phase("cand_gather"):
                _comm.all_gather(out, send)
        if side is not None:
            candidates.record_stream(side)
            cur = torch.cuda.current_stream()
            out.record_stream(cur)                  # allocated under the side stream, consumed on this one
            if self._valid is not None:
                self._valid.record_stream(cur)
        self._out = out

    def cancel(self):
        """Drop a gather that was started for a batch nobody will train on (the collective itself still runs to
        completion on the side stream: every rank issued it)."""
        self._out = None
        self._valid = None

    def wait(self) -> tp.Tuple[torch.Tensor, int, tp.Optional[torch.Tensor]]:
        out, valid = self._out, self._valid
        self._out = self._valid = None
        if out is None:
            raise RuntimeError("CandidateGather.wait() without a matching start()")
        if not is_distributed():
            return out, 0, None
        if self.stream is not None and out.is_cuda:
            with _phase("gather_wait"):
                torch.cuda.current_stream().wait_stream(self.stream)
        return out, rank() * self._B, valid


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
        _comm.all_gather(out, candidates)
        return out

    @staticmethod
    def backward(ctx, grad):
        world = world_size()
        grad = grad.contiguous()
        B = grad.shape[0] // world
        out = torch.empty((B,) + tuple(grad.shape[1:]), device=grad.device, dtype=grad.dtype)
        _comm.reduce_scatter(out, grad)
        return out


def gather_learnable_candidates(candidates: torch.Tensor):
    """-> (gathered [world*B, ...], target_offset); identity at world_size 1."""
    if not is_distributed():
        return candidates, 0
    return GatherCandidatesFn.apply(candidates), rank() * candidates.shape[0]
