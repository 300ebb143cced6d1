"""In-process loopback communicator: N replicas of the data-parallel step on ONE GPU (test infrastructure).

The GPU box of the test tier has a single device, so the N > 1 code of ``brainmagick_amd.distrib`` (candidate
all-gather with ``target_offset``, in-place reduce-scatter of the flat gradient bucket, sharded Adam, in-place
all-gather of the parameters, buffer averaging, moment gather for checkpoints) can never meet a second rank there.
``LoopbackComm`` implements the communicator interface of ``distrib._RcclComm`` between N Python threads of one
process: every replica runs the unmodified ``Solver.train_step`` in its own thread; the threads hand a baton
around so that exactly one of them runs at any time (the library's host-side caches are not re-entrant, and
this makes the run deterministic) and meet at every collective.

Data moves with plain device copies enqueued on the calling replica's current stream after an event of the
depositor's stream, so the GPU-side ordering is that of a real communicator.
"""
import threading
import typing as tp

import torch

from brainmagick_amd import distrib


class LoopbackComm:
    kind = "loopback"

    def __init__(self, world: int):
        self.world = world
        self._tls = threading.local()
        self._baton = threading.Lock()
        self._barrier = threading.Barrier(world)
        self._slots: tp.List[tp.Any] = [None] * world
        self.calls: tp.List[str] = []              # collective names in issue order (rank 0's view)
        self._gpu = torch.cuda.is_available()

    # -- per-thread identity -----------------------------------------------------------------------
    @property
    def rank(self) -> int:
        return self._tls.rank

    def enter(self, rank: int):
        self._tls.rank = rank
        self._baton.acquire()

    def leave(self):
        self._baton.release()

    # -- rendezvous ----------------------------------------------------------------------------------
    def _meet(self, name: str, payload, move):
        """Deposit `payload`, wait for every rank, run `move(payloads)` holding the baton, wait again (nobody
        re-deposits or touches a deposited tensor before every rank has moved its data)."""
        r = self.rank
        if r == 0:
            self.calls.append(name)
        ev = None
        if self._gpu:
            ev = torch.cuda.Event()
            ev.record()
        self._slots[r] = (name, payload, ev)
        self._baton.release()
        try:
            self._barrier.wait(timeout=300)
        finally:
            self._baton.acquire()
        names = {s[0] for s in self._slots}
        assert len(names) == 1, f"ranks disagree on the collective: {names}"
        if self._gpu:
            for s in self._slots:
                torch.cuda.current_stream().wait_event(s[2])
        move([s[1] for s in self._slots])
        self._baton.release()
        try:
            self._barrier.wait(timeout=300)
        finally:
            self._baton.acquire()

    def abort(self):
        self._barrier.abort()

    # -- communicator interface (distrib._RcclComm) -----------------------------------------------------
    def all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        n = inp.numel()

        def move(inps):
            for r, t in enumerate(inps):
                out.view(-1)[r * n:(r + 1) * n].copy_(t.reshape(-1))
        self._meet("all_gather", inp, move)

    def reduce_scatter_shard(self, flat: torch.Tensor):
        lo, hi = distrib.shard_bounds(flat.numel(), self.world, self.rank)
        mine = self.rank

        def move(flats):
            # fixed order of the sum (rank 0 first): the same bits on every run
            acc = flats[0][lo:hi].clone()
            for t in flats[1:]:
                acc += t[lo:hi]
            flats[mine][lo:hi].copy_(acc)
        self._meet("reduce_scatter_shard", flat, move)

    def all_gather_shards(self, flat: torch.Tensor):
        mine = self.rank

        def move(flats):
            for r, t in enumerate(flats):
                if r != mine:
                    lo, hi = distrib.shard_bounds(flat.numel(), self.world, r)
                    flat[lo:hi].copy_(t[lo:hi])
        self._meet("all_gather_shards", flat, move)

    def reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor):
        n = out.numel()
        mine = self.rank

        def move(inps):
            acc = inps[0].reshape(-1)[mine * n:(mine + 1) * n].clone()
            for t in inps[1:]:
                acc += t.reshape(-1)[mine * n:(mine + 1) * n]
            out.view(-1).copy_(acc)
        self._meet("reduce_scatter", inp, move)

    def all_reduce(self, t: torch.Tensor, op: str = "sum"):
        def move(copies):
            acc = copies[0].clone()
            for c in copies[1:]:
                acc = acc + c if op == "sum" else torch.maximum(acc, c)
            t.copy_(acc)
        self._meet("all_reduce", t.clone(), move)

    def barrier(self):
        self._meet("barrier", None, lambda _: None)

    def scalar_device(self):
        return "cuda" if self._gpu else "cpu"

    def close(self):
        pass


def run_replicas(world: int, body: tp.Callable[[int], tp.Any]) -> tp.List[tp.Any]:
    """Run ``body(rank)`` in `world` threads under one LoopbackComm installed as the process' communicator;
    returns the per-rank results, re-raises the first failure."""
    comm = LoopbackComm(world)
    prev = distrib._comm
    distrib._comm = comm
    results: tp.List[tp.Any] = [None] * world
    errors: tp.List[tp.Optional[BaseException]] = [None] * world

    def runner(r):
        comm.enter(r)
        try:
            if comm._gpu:
                torch.cuda.set_device(0)
            # backward on the CALLING thread: with the engine's shared per-device worker thread a collective inside a
            # backward node (the adjoint of the learnable-candidate gather) would block the other replica's nodes
            with torch.autograd.set_multithreading_enabled(False):
                results[r] = body(r)
        except BaseException as exc:            # noqa: BLE001 -- reported by the caller
            errors[r] = exc
            comm.abort()
        finally:
            comm.leave()

    threads = [threading.Thread(target=runner, args=(r,), daemon=True) for r in range(world)]
    try:
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=900)
    finally:
        distrib._comm = prev
    for exc in errors:
        if exc is not None and not isinstance(exc, threading.BrokenBarrierError):
            raise exc
    for exc in errors:
        if exc is not None:
            raise exc
    return results
