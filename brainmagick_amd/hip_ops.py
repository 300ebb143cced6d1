"""Typed Python wrappers over the libbmhip C-ABI (one wrapper per entry point group).

PyTorch is used here only as the owner of device memory and of the HIP stream: every function
checks that its tensors are fp32 / contiguous / on the GPU, allocates outputs with ``torch.empty``
and enqueues the HIP kernels on the current stream.  CPU tensors are rejected -- there is no
fallback path.
"""
import ctypes
import typing as tp

import torch

from ._lib import lib, check, BmHipError

ACT_NONE, ACT_GELU, ACT_RELU, ACT_LEAKY = 0, 1, 2, 3
BKC = 16

# Compute mode of the MFMA contractions.  Activations / parameters / gradients stay fp32 in HBM.
#   "f16x2" (default) fp32-ACCURATE contraction on the f16 matrix cores: every operand is scaled by a power of two (per
#           tensor for activations, per output row for weights) and split into TWO f16 planes, three partial
#           products per block, fp32 accumulate, exact inverse scaling; half the matrix-core work of "f32x3",
#           same parity tolerances (norm-wise error bound, see csrc/conv_nn_h2w.hip).  Shapes the wide f16x2
#           kernels do not cover run on the "f32x3" kernels;
#   "f32x3" fp32-ACCURATE contraction on the bf16 matrix cores: both fp32 operands are split
#           EXACTLY into three bf16 planes, six partial products per block, fp32 accumulate; measured
#           error vs fp64 <= the exact-fp32 MFMA path's, held to the same parity tolerances;
#   "f32"   exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), bit-identical to an fp32 FMA chain.
# (Round 4 removed the opt-in reduced-precision "bf16" mode and the wide-tile f32x3 kernels: every mode left is
# fp32-class and held to the same tolerances.)
import os as _os
DEFAULT_COMPUTE_DTYPE = "f16x2"
COMPUTE_DTYPES = ("f32", "f32x3", "f16x2")
_compute_dtype = _os.environ.get("BM_COMPUTE_DTYPE", DEFAULT_COMPUTE_DTYPE)
if _compute_dtype not in COMPUTE_DTYPES:
    raise ValueError(f"BM_COMPUTE_DTYPE must be one of {COMPUTE_DTYPES}, got {_compute_dtype!r}")


def set_compute_dtype(name: str):
    global _compute_dtype
    if name not in COMPUTE_DTYPES:
        raise ValueError(f"compute dtype must be one of {COMPUTE_DTYPES}, got {name!r}")
    _compute_dtype = name


def get_compute_dtype() -> str:
    return _compute_dtype


class KernelTimer:
    """Optional HIP-event timing of the MFMA kernels, on the stream they are launched on (the
    current torch stream).  ``bench.py`` installs one to measure the dominant kernel's average
    launch duration live inside the timed region; cost = two event records per launch."""

    def __init__(self):
        self.records: tp.List[tp.Tuple[str, float, torch.cuda.Event, torch.cuda.Event]] = []

    def launch(self, name: str, flops: float, fn):
        start = torch.cuda.Event(enable_timing=True)
        end = torch.cuda.Event(enable_timing=True)
        start.record()
        out = fn()
        end.record()
        self.records.append((name, flops, start, end))
        return out

    def summary(self) -> tp.Dict[str, tp.Dict[str, float]]:
        """name -> {launches, avg_ms, median_ms, max_ms, outliers, flops_per_launch}; call after a device
        synchronize.  ``avg_ms`` is the mean over the launches whose duration is within 20x the label's median: an
        event pair also measures whatever keeps the queue from reaching the kernel (a host stall between the two
        records shows up as a launch of many milliseconds); the dropped samples are counted in ``outliers``."""
        samples: tp.Dict[str, tp.List[float]] = {}
        flops_sum: tp.Dict[str, float] = {}
        for name, flops, start, end in self.records:
            samples.setdefault(name, []).append(start.elapsed_time(end))
            flops_sum[name] = flops_sum.get(name, 0.0) + flops
        out = {}
        for name, v in samples.items():
            med = sorted(v)[len(v) // 2]
            kept = [t for t in v if t <= 20.0 * med]
            out[name] = dict(launches=len(v), avg_ms=sum(kept) / len(kept), median_ms=med, max_ms=max(v),
                             outliers=len(v) - len(kept), flops_per_launch=flops_sum[name] / len(v))
        return out


_timer: tp.Optional[KernelTimer] = None


def set_kernel_timer(timer: tp.Optional[KernelTimer]):
    global _timer
    _timer = timer


def _p(t: tp.Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t: torch.Tensor, name: str, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise BmHipError(f"{name}: expected a GPU tensor; the brainmagick_amd hot path has no CPU "
                         "fallback (got %s)" % (t.device if isinstance(t, torch.Tensor) else type(t)))
    if t.dtype != dtype:
        raise BmHipError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise BmHipError(f"{name}: expected a contiguous tensor")
    return t


def _opt(t, name, dtype=torch.float32):
    return None if t is None else _req(t, name, dtype)


# ------------------------------------------------------------------------------------------------
def conv_mpad(M: int) -> int:
    return lib().bm_conv_mpad(M)


AMAX_SHARDS = 8       # an amax slot is 8 floats (csrc/bm_common.h; the finalize kernels write all eight); max|x| = slot.max()
amax_scans = 0        # number of stand-alone amax passes launched (producers that publish their own maximum need none)


def amax(x: torch.Tensor, nonfinite_flag: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
    """[AMAX_SHARDS] fp32 tensor whose maximum is max |x| (device side, no sync).  Cached on the tensor object together with
    its version counter, so a tensor consumed by several contractions (forward conv, weight gradient) is
    scanned once and an in-place modification invalidates the cache.  ``nonfinite_flag`` (int32 device tensor):
    element 0 is set to 1 by the same pass if x holds an inf / nan (a cached tensor was checked when it was scanned)."""
    cached = getattr(x, "_bm_amax", None)
    if cached is not None and cached[0] == x._version and cached[1] == x.data_ptr() and \
            (nonfinite_flag is None or cached[3]):
        return cached[2]
    _req(x, "amax.x")
    global amax_scans
    amax_scans += 1
    out = torch.empty(AMAX_SHARDS, device=x.device, dtype=torch.float32)
    check(lib().bm_amax_checked(_p(x), x.numel(), _p(out), _p(_amax_ws(x.device)),
                                _p(_opt(nonfinite_flag, "nonfinite_flag", torch.int32)), _stream()), "bm_amax")
    try:
        x._bm_amax = (x._version, x.data_ptr(), out, nonfinite_flag is not None)
    except Exception:       # tensors that refuse attributes: just do not cache
        pass
    return out


_amax_pool: tp.Dict[torch.device, tp.List[tp.Any]] = {}
_amax_workspaces: tp.Dict[tp.Any, torch.Tensor] = {}


def _amax_ws(device) -> torch.Tensor:
    """Scratch for the per-workgroup partial maxima: one buffer per (device, stream); kernels of one stream run in
    order, so every producer of that stream can use the same buffer."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _amax_workspaces.get(key)
    if ws is None:
        ws = _amax_workspaces[key] = torch.empty(lib().bm_amax_ws_elems(), device=device, dtype=torch.float32)
    return ws


def _amax_slot(t: torch.Tensor) -> tp.Optional[torch.Tensor]:
    """An [AMAX_SHARDS] fp32 slot for a producer kernel to publish max|t| into (f16x2 mode only; None otherwise).
    Slots are carved out of pooled buffers, each slot is used once.  The slot is attached to the tensor like
    `amax()` would, so the consuming contraction finds it without a pass over the tensor."""
    if _compute_dtype != "f16x2" or t.numel() == 0:      # empty: the producer returns before it writes
        return None
    pool = _amax_pool.get(t.device)
    if pool is None or pool[1] >= pool[0].numel():
        # one zero fill per 4 096 slots (a slot nobody finalized reads as max|x| = 0)
        pool = _amax_pool[t.device] = [torch.zeros(4096 * AMAX_SHARDS, device=t.device, dtype=torch.float32), 0]
    slot = pool[0][pool[1]:pool[1] + AMAX_SHARDS]
    pool[1] += AMAX_SHARDS
    t._bm_amax = (t._version, t.data_ptr(), slot, False)
    return slot


def _row_amax_out(t: torch.Tensor, slot) -> tp.Optional[torch.Tensor]:
    """[channels] fp32 buffer for a producer with a (channel, split) grid to publish max|t| PER CHANNEL into (next to the
    tensor slot).  A gradient channel is a row of the weight gradient: the f16x2 weight-gradient
    kernel then scales A row by row (csrc/gemm_nt_h2w.hip, RS kernels), so a channel far below its tensor's maximum keeps
    its 22 bits.  Attached to the tensor like the slot."""
    if slot is None:
        return None
    rows = torch.empty(t.shape[1], device=t.device, dtype=torch.float32)
    t._bm_row_amax = (t._version, t.data_ptr(), rows)
    return rows


def row_amax_of(t: torch.Tensor) -> tp.Optional[torch.Tensor]:
    cached = getattr(t, "_bm_row_amax", None)
    if cached is not None and cached[0] == t._version and cached[1] == t.data_ptr():
        return cached[2]
    return None


def _slot_args(slot):
    """(amax_out, amax_ws) arguments of a producer."""
    return (_p(slot), _p(_amax_ws(slot.device)) if slot is not None else None)


def _touched(t: torch.Tensor) -> torch.Tensor:
    """Called by every wrapper that lets a library kernel write INTO an existing tensor through its raw pointer:
    torch's version counter does not see such writes, so a maximum published for the old contents must go."""
    for attr in ("_bm_amax", "_bm_row_amax", "_bm_inv_norms"):
        if getattr(t, attr, None) is not None:
            try:
                delattr(t, attr)
            except AttributeError:
                pass
    return t


def share_amax(src: torch.Tensor, view: torch.Tensor) -> torch.Tensor:
    """``view`` is a slice of ``src``: the maximum of the whole tensor bounds the slice (any upper bound within the
    f16 headroom serves as the scale), so the slice needs no pass of its own."""
    if _compute_dtype != "f16x2":     # nobody consumes a maximum in the other compute modes: no scan
        return view
    slot = amax(src)
    try:
        view._bm_amax = (view._version, view.data_ptr(), slot, False)
    except Exception:
        pass
    return view


# ------------------------------------------------------------------------------------------------
# Packed parameters.  A model's weights change once per optimizer step, so in "f16x2" mode the packed form of
# every PARAMETER (leaf tensor that requires grad) that the convs ask for lives in a persistent buffer, and all
# of them are refreshed together by one launch (bm_pack_weights_h2_batch) the first time one is asked for after
# the parameters changed -- instead of one small launch per conv, forward and backward.  "Changed" = the
# library's own writers said so (weights_changed(): FlatAdam.step, the data-parallel gathers) or the tensor's
# autograd version moved (load_state_dict, in-place edits); evaluation never re-packs.
_weights_epoch = 0
pack_launches = 0            # launches of either packing kernel (tests / bench bookkeeping)


def weights_changed():
    """Called by every library routine that writes parameters through raw pointers."""
    global _weights_epoch
    _weights_epoch += 1


class _PackPlan:
    MAX_JOBS = 512           # e.g. many models in one test session: start over rather than grow without bound
    KEEP = 3                 # parameter updates an entry survives without being asked for (a model that went away)

    def __init__(self, device):
        self.device = device
        self.entries: tp.Dict[tuple, dict] = {}
        self.table: tp.Optional[torch.Tensor] = None
        self.total_blocks = 0

    def get(self, src: torch.Tensor, geom: tuple) -> torch.Tensor:
        key = (src.data_ptr(),) + geom
        e = self.entries.get(key)
        if e is None:
            if len(self.entries) >= self.MAX_JOBS:
                self.entries.clear()
            G, M, Cin, KS = geom[:4]
            dst = torch.empty(lib().bm_packed_weight_bytes_h2(G, M, Cin, KS), device=src.device, dtype=torch.uint8)
            dst._bm_mode = "f16x2"
            dst._bm_groups = G
            # `src` is kept alive: the batched launch reads it through its raw pointer
            e = self.entries[key] = dict(src=src, ptr=src.data_ptr(), geom=geom, dst=dst, stamp=None)
            self.table = None
        e["used"] = _weights_epoch
        if e["stamp"] != (_weights_epoch, src._version):
            self.refresh()
        return e["dst"]

    def refresh(self):
        global pack_launches
        # parameters nobody asked for lately (another model of the process, a re-seated `.data`) leave the plan
        stale = [k for k, e in self.entries.items()
                 if e["used"] < _weights_epoch - self.KEEP or e["src"].data_ptr() != e["ptr"]]
        for k in stale:
            del self.entries[k]
        if stale:
            self.table = None
        ents = list(self.entries.values())
        if not ents:
            return
        if self.table is None:
            nb = lib().bm_pack_h2_job_bytes()
            host = ctypes.create_string_buffer(nb * len(ents))
            block0 = 0
            for i, e in enumerate(ents):
                n = lib().bm_pack_h2_job_fill(ctypes.c_void_p(ctypes.addressof(host) + i * nb), _p(e["src"]),
                                              _p(e["dst"]), *e["geom"], None, block0)
                if n < 0:
                    raise BmHipError("bm_pack_h2_job_fill: bad arguments %r" % (e["geom"],))
                block0 += n
            self.table = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(self.device)
            self.total_blocks = block0
        max_nk = max(e["geom"][2] * e["geom"][3] for e in ents)          # Cin * KS
        check(lib().bm_pack_weights_h2_batch(_p(self.table), len(ents), self.total_blocks, max_nk, _stream()),
              "bm_pack_weights_h2_batch")
        pack_launches += 1
        for e in ents:
            e["stamp"] = (_weights_epoch, e["src"]._version)


_pack_plans: tp.Dict[tp.Tuple[str, int], _PackPlan] = {}


def _pack_plan(device) -> _PackPlan:
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _pack_plans:
        _pack_plans[key] = _PackPlan(torch.device(*key))
    return _pack_plans[key]


def pack_weights(src: torch.Tensor, G: int, M: int, Cin: int, KS: int, sg: int, sm: int, sc: int,
                 sj: int, flip: bool = False, alpha: tp.Optional[torch.Tensor] = None,
                 shape: tp.Optional[tp.Tuple[int, int]] = None) -> torch.Tensor:
    """``shape`` = (T, dilation) of the conv that will consume the packed weights: in "f16x2" mode it decides
    between the wide f16x2 kernel's layout and the 3 x bf16 layout of the narrow kernels."""
    _req(src, "pack_weights.src")
    mode = _compute_dtype
    if mode == "f16x2":
        if shape is not None and lib().bm_conv_h2_covers(Cin, M, shape[0], KS, shape[1]):
            if alpha is None and src.requires_grad and src.is_leaf:
                return _pack_plan(src.device).get(src, (G, M, Cin, KS, sg, sm, sc, sj, int(flip)))
            nbytes = lib().bm_packed_weight_bytes_h2(G, M, Cin, KS)
            dst = torch.empty(nbytes, device=src.device, dtype=torch.uint8)
            check(lib().bm_pack_weights_h2(_p(src), _p(dst), G, M, Cin, KS, sg, sm, sc, sj, int(flip),
                                           _p(_opt(alpha, "alpha")), _stream()), "bm_pack_weights_h2")
            global pack_launches
            pack_launches += 1
            dst._bm_mode = "f16x2"
            dst._bm_groups = G
            return dst
        mode = "f32x3"
    if mode == "f32x3":
        n = lib().bm_packed_weight_elems_x3(G, M, Cin, KS)
        dst = torch.empty(n, device=src.device, dtype=torch.bfloat16)
        check(lib().bm_pack_weights_x3(_p(src), _p(dst), G, M, Cin, KS, sg, sm, sc, sj, int(flip),
                                       _p(_opt(alpha, "alpha")), _stream()), "bm_pack_weights_x3")
        dst._bm_mode = mode                      # tells conv_nn which kernel family packed it
        return dst
    n = lib().bm_packed_weight_elems(G, M, Cin, KS)
    dst = torch.empty(n, device=src.device, dtype=torch.float32)
    check(lib().bm_pack_weights(_p(src), _p(dst), G, M, Cin, KS, sg, sm, sc, sj, int(flip),
                                _p(_opt(alpha, "alpha")), _stream()), "bm_pack_weights")
    return dst


def pack_conv_fwd(weight: torch.Tensor, shape=None) -> torch.Tensor:
    """nn.Conv1d weight [M, Cin, KS] for the forward conv."""
    M, Cin, KS = weight.shape
    return pack_weights(weight, 1, M, Cin, KS, 0, Cin * KS, KS, 1, shape=shape)


def pack_conv_dgrad(weight: torch.Tensor, shape=None) -> torch.Tensor:
    """nn.Conv1d weight [M, Cin, KS] for the data gradient: roles of M/Cin swapped, taps flipped."""
    M, Cin, KS = weight.shape
    return pack_weights(weight, 1, Cin, M, KS, 0, KS, Cin * KS, 1, flip=True, shape=shape)


def conv_nn(x: torch.Tensor, wpacked: torch.Tensor, M: int, KS: int = 1, dil: int = 1,
            widx: tp.Optional[torch.Tensor] = None, bias=None, scale=None, shift=None, res=None,
            act: int = ACT_NONE, leak: float = 0., want_pre: bool = False, want_out: bool = True,
            want_stats: bool = False, out: tp.Optional[torch.Tensor] = None, bias_gstride: int = 0,
            publish_amax: bool = True):
    """Returns (y_pre | None, y_out | None, stats | None); x is [B, Cin, T].  ``out``: write y_out there instead of
    allocating (it may be ``res`` itself: every element is read and written by the same thread, which is how the
    ClipLoss backward accumulates over candidate blocks).  ``bias_gstride`` > 0: ``bias`` holds one vector per weight
    group, ``bias_gstride`` floats apart, selected by ``widx`` like the weights.  ``publish_amax=False``: the output
    goes to an elementwise kernel (the data gradients of the conv stack), nobody needs its maximum."""
    _req(x, "conv_nn.x")
    mode = getattr(wpacked, "_bm_mode", "f32")      # set by pack_weights
    _req(wpacked, "conv_nn.w", {"f32": torch.float32, "f16x2": torch.uint8}.get(mode, torch.bfloat16))
    B, Cin, T = x.shape
    y_pre = torch.empty(B, M, T, device=x.device, dtype=torch.float32) if want_pre else None
    if out is not None:
        _req(out, "conv_nn.out")
        assert want_out and out.numel() == B * M * T, (out.shape, (B, M, T))
        y_out = out
        _touched(y_out)
    else:
        y_out = torch.empty(B, M, T, device=x.device, dtype=torch.float32) if want_out else None
    stats = None
    if want_stats:
        if mode == "f16x2":     # channel-major: a channel's partials are one contiguous run for bn_finalize
            stats = torch.empty(M, lib().bm_conv_h2_stats_tiles(B, T), 2, device=x.device, dtype=torch.float32)
            stats._bm_channel_major = True
        else:
            stats = torch.empty(lib().bm_conv_stats_tiles(B, T), M, 2, device=x.device, dtype=torch.float32)
    if res is not None:
        _req(res, "conv_nn.res")
        assert res.shape == (B, M, T), (res.shape, (B, M, T))
    common = (_p(_opt(widx, "widx", torch.int32)), _p(_opt(bias, "bias")), bias_gstride, _p(_opt(scale, "scale")),
              _p(_opt(shift, "shift")), _p(res), M * T, _p(y_pre), _p(y_out), M * T, _p(stats), B, Cin, M, T,
              KS, dil, act, leak)
    if mode == "f16x2":
        x_amax = amax(x)
        y_slot = _amax_slot(y_out) if (y_out is not None and publish_amax) else None

        def launch():
            check(lib().bm_conv1d_nn_h2(_p(x), Cin * T, _p(x_amax), _p(wpacked), *common, wpacked._bm_groups,
                                        *_slot_args(y_slot), _stream()), "bm_conv1d_nn_h2")
    else:
        fn = {"f32": lib().bm_conv1d_nn, "f32x3": lib().bm_conv1d_nn_x3}[mode]

        def launch():
            check(fn(_p(x), Cin * T, _p(wpacked), *common, _stream()), f"bm_conv1d_nn[{mode}]")
    if _timer is not None:
        if mode == "f16x2":
            label = f"conv_nn_h2w_kernel<{KS},{lib().bm_conv_h2_mw_for(M)}>"
        else:
            label = {"f32": f"conv_nn_kernel<{lib().bm_conv_mt_for(M)}>",
                     "f32x3": f"conv_nn_x3_kernel<{lib().bm_conv_x3_mt_for(M)}>"}[mode]
        _timer.launch(label, 2.0 * B * T * M * Cin * KS, launch)
    else:
        launch()
    return y_pre, y_out, stats


# Device-side "index out of range" flag (one int32 per device).  The grouped kernels never read outside
# their weight tables (bad indices are clamped to group 0 by bm_index_to_i32 / skipped by
# bm_group_by_index); the flag is raised as an IndexError at the next synchronisation point the caller
# chooses (`raise_if_index_error`, called by Solver next to the reference's isfinite asserts), or
# immediately with BM_CHECK_INDICES=1.
_index_err: tp.Dict[torch.device, torch.Tensor] = {}
_CHECK_INDICES_NOW = _os.environ.get("BM_CHECK_INDICES", "0") == "1"


def index_error_flag(device) -> torch.Tensor:
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    flag = _index_err.get(device)
    if flag is None:
        # [index out of range, non-finite input, unsupported (not all-true) ClipLoss mask]
        flag = torch.zeros(3, device=device, dtype=torch.int32)
        _index_err[device] = flag
    return flag


def raise_if_index_error(device=None):
    """Synchronising check of the flag; raises like the reference's out-of-range gather would."""
    flags = list(_index_err.values()) if device is None else [index_error_flag(device)]
    for flag in flags:
        if int(flag[0].item()) != 0:
            flag[0:1].zero_()       # (the other two words -- non-finite input, bad mask -- belong to the Solver)
            raise IndexError("subject / layout index out of range for the weight table "
                             "(bm/models/common.py:57 would raise in `weights.gather`)")


def index_i32(idx: torch.Tensor, G: int) -> torch.Tensor:
    """idx [B] int64 -> int32 group indices, range-checked against [0, G) on the device."""
    _req(idx, "index_i32.idx", torch.int64)
    out = torch.empty(idx.numel(), device=idx.device, dtype=torch.int32)
    check(lib().bm_index_to_i32(_p(idx), idx.numel(), G, _p(out), _p(index_error_flag(idx.device)),
                                _stream()), "bm_index_to_i32")
    if _CHECK_INDICES_NOW:
        raise_if_index_error(idx.device)
    return out


def group_by_index(idx: torch.Tensor, G: int):
    """idx [B] int64 -> (order [B] int32, seg [G+1] int32)."""
    _req(idx, "group_by_index.idx", torch.int64)
    B = idx.numel()
    order = torch.empty(B, device=idx.device, dtype=torch.int32)
    seg = torch.empty(G + 1, device=idx.device, dtype=torch.int32)
    check(lib().bm_group_by_index(_p(idx), B, G, _p(order), _p(seg), _p(index_error_flag(idx.device)),
                                  _stream()), "bm_group_by_index")
    return order, seg


def gemm_nt(a: torch.Tensor, x: torch.Tensor, S: int, M: int, Cn: int, T: int, KS: int = 1,
            dil: int = 1, a_strides=None, x_strides=None, order=None, seg=None, G: int = 1,
            out: tp.Optional[torch.Tensor] = None, out_strides=None, nsplit: tp.Optional[int] = None,
            force_f32: bool = False):
    """out[g*sg + m*sm + c*sc + j*sj] = sum_{s in g} sum_t a[s][m][t] * x[s][c][t + shift_j].

    (``force_f32`` is kept in the signature for callers whose result feeds a softmax: every mode left is fp32-class.)

    a_strides / x_strides = (segment stride, row stride) in elements; defaults are contiguous
    [S][rows][T].  Default ``out`` is [G][M][Cn][KS] contiguous."""
    _req(a, "gemm_nt.a")
    _req(x, "gemm_nt.x")
    if a_strides is None:
        a_strides = (M * T, T)
    if x_strides is None:
        x_strides = (Cn * T, T)
    if out_strides is None:
        out_strides = (M * Cn * KS, Cn * KS, KS, 1)
    if out is None:
        out = torch.empty(G, M, Cn, KS, device=a.device, dtype=torch.float32)
    mode = _compute_dtype
    grouped = order is not None or seg is not None
    if mode == "f16x2":
        # rows T apart; the segment stride only matters when there is more than one segment
        contiguous = a_strides[1] == T and x_strides[1] == T and \
            (S == 1 or (a_strides[0] == M * T and x_strides[0] == Cn * T))
        covered = contiguous and lib().bm_gemm_nt_h2_covers(M, Cn, KS, S, T, G, dil, int(grouped))
        if contiguous and not covered and KS == 1 and lib().bm_gemm_nt_h2_covers(Cn, M, KS, S, T, G, dil, int(grouped)):
            # sum_t a[m][t] x[c][t] is symmetric in its operands: with the roles swapped the (M, Cn) rectangle may fit
            # the wide tiles (208 x 270 per (layout, subject) pair: 256 x 128 tiles) where (270, 208) does not --
            # the result lands through the swapped output strides
            a, x, M, Cn, a_strides, x_strides = x, a, Cn, M, x_strides, a_strides
            out_strides = (out_strides[0], out_strides[2], out_strides[1], out_strides[3])
            covered = True
        if not covered:
            mode = "f32x3"      # shapes without a wide f16x2 kernel: the (equally fp32-accurate) 3 x bf16 kernels
        elif nsplit is None:
            nsplit = lib().bm_gemm_nt_h2_suggest_splits_grouped(M, Cn, KS, S, T, G)
    if nsplit is None:
        nsplit = lib().bm_gemm_nt_suggest_splits(M, Cn, KS, S, T, G)
        if G > 1:
            # grouped (per-subject / per-layout) weight gradients run in the 160 x 128-tile kernels, two workgroups
            # per CU: split each group's (segment, 32-sample chunk) list until ~512 workgroups exist, keeping at
            # least 8 chunks per split
            tiles = -(-M // 160) * -(-Cn // 128) * G
            chunks = max(1, S // G) * -(-T // 32)
            nsplit = max(1, min(8, 512 // tiles, chunks // 8))
    canonical = tuple(out_strides) == (M * Cn * KS, Cn * KS, KS, 1)
    if nsplit == 1 and canonical:
        part = out
    else:
        part = torch.empty(G * nsplit * M * Cn * KS, device=a.device, dtype=torch.float32)
    if mode == "f16x2":
        a_amax, x_amax = amax(a), amax(x)
        a_rows = row_amax_of(a)          # published by the producer of `a` (act_bn_bwd / glu_bwd)
        if a_rows is not None and a_rows.numel() != M:
            a_rows = None

        def launch():
            if grouped:
                check(lib().bm_gemm_nt_h2_grouped(_p(a), a_strides[0], a_strides[1], _p(a_amax), _p(x), x_strides[0],
                                                  x_strides[1], _p(x_amax), _p(_opt(order, "order", torch.int32)),
                                                  _p(_opt(seg, "seg", torch.int32)), _p(part), S, G, M, Cn, T, nsplit,
                                                  _stream()), "bm_gemm_nt_h2_grouped")
                return
            check(lib().bm_gemm_nt_h2_rows(_p(a), a_strides[0], a_strides[1], _p(a_amax), _p(a_rows), _p(x),
                                           x_strides[0], x_strides[1], _p(x_amax), _p(part), S, M, Cn, T, KS, dil,
                                           nsplit, _stream()), "bm_gemm_nt_h2")
    else:
        fn = {"f32": lib().bm_gemm_nt, "f32x3": lib().bm_gemm_nt_x3}[mode]

        def launch():
            check(fn(_p(a), a_strides[0], a_strides[1], _p(x), x_strides[0], x_strides[1],
                     _p(_opt(order, "order", torch.int32)), _p(_opt(seg, "seg", torch.int32)), _p(part), S,
                     G, M, Cn, T, KS, dil, nsplit, _stream()), "bm_gemm_nt")
    if _timer is not None:
        suffix = {"f32": "", "f32x3": "_x3", "f16x2": "_h2w"}[mode]
        _timer.launch(f"gemm_nt{suffix}_kernel<KS={KS}>", 2.0 * S * T * M * Cn * KS, launch)
    else:
        launch()
    if part is not out:
        check(lib().bm_reduce_splits(_p(part), _p(out), G, nsplit, M, Cn, KS, *out_strides,
                                     _stream()), "bm_reduce_splits")
    return out


class _DirectArmed:
    """(a plain counter, not thread-local: the autograd engine runs the backward nodes on its own worker thread)"""
    depth = 0

    @property
    def on(self):
        return self.depth > 0


_direct_armed = _DirectArmed()


class direct_grads_armed:
    """Context in which ``grad_destination`` hands out the flat-bucket views (``FlatAdam.writing_grads``): the
    backward pass of the training step, and nothing else -- gradients returned by ``torch.autograd.grad`` or produced
    by a stray backward must not alias the optimizer's bucket."""

    def __enter__(self):
        _direct_armed.depth += 1
        return self

    def __exit__(self, *exc):
        _direct_armed.depth -= 1
        return False


def grad_destination(param: torch.Tensor) -> tp.Optional[torch.Tensor]:
    """Where the weight gradient of ``param`` may be written directly: its view of the optimizer's flat gradient bucket
    (``FlatAdam`` registers it as ``param._bm_grad_dst``), handed out ONCE per ``zero_grad`` -- a parameter that
    takes part in the graph twice gets a fresh tensor the second time, and autograd accumulates as usual.  The
    returned view becomes ``param.grad`` without a copy (``FlatAdam.collect_grads`` recognises the address)."""
    dst = getattr(param, "_bm_grad_dst", None)
    if dst is None or dst[1][0] or param.grad is not None or not _direct_armed.on:
        return None
    dst[1][0] = True
    return dst[0]


def gemm_nt_partials(a, x, S, M, Cn, T, a_strides, x_strides, nsplit=None):
    """Split-K partial tiles [nsplit][M][Cn] (KS=1, one group), consumed by clip_ce."""
    _req(a, "gemm_nt.a")
    _req(x, "gemm_nt.x")
    mode = _compute_dtype
    if mode == "f16x2" and not lib().bm_gemm_nt_h2_covers(M, Cn, 1, S, T, 1, 1, 0):
        mode = "f32x3"
    # the score contraction proper (dense [M][T] x [Cn][T]): 256 x 256 tiles, transposed vector stores
    scores_kernel = mode == "f16x2" and S == 1 and tuple(a_strides) == (0, T) and tuple(x_strides) == (0, T) and \
        bool(lib().bm_clip_scores_h2_covers(M, Cn, T))
    if nsplit is None:
        if scores_kernel:
            nsplit = lib().bm_clip_scores_h2_suggest_splits(M, Cn, T)
        else:
            nsplit = lib().bm_gemm_nt_h2_suggest_splits(M, Cn, 1, S, T) if mode == "f16x2" else \
                lib().bm_clip_suggest_splits(M, Cn, S, T)
    part = torch.empty(nsplit, M, Cn, device=a.device, dtype=torch.float32)
    if scores_kernel:
        a_amax, x_amax = amax(a), amax(x)

        def launch():
            check(lib().bm_clip_scores_h2(_p(a), _p(a_amax), _p(x), _p(x_amax), _p(part), M, Cn, T, nsplit, _stream()),
                  "bm_clip_scores_h2")
    elif mode == "f16x2":
        a_amax, x_amax = amax(a), amax(x)

        def launch():
            check(lib().bm_gemm_nt_h2(_p(a), a_strides[0], a_strides[1], _p(a_amax), _p(x), x_strides[0],
                                      x_strides[1], _p(x_amax), _p(part), S, M, Cn, T, 1, 1, nsplit, _stream()),
                  "bm_gemm_nt_h2")
    else:
        fn = lib().bm_gemm_nt_x3 if mode == "f32x3" else lib().bm_gemm_nt

        def launch():
            check(fn(_p(a), a_strides[0], a_strides[1], _p(x), x_strides[0], x_strides[1], None, None, _p(part),
                     S, 1, M, Cn, T, 1, 1, nsplit, _stream()), "bm_gemm_nt")
    if _timer is not None:
        _timer.launch("clip_scores:gemm_nt" + {"f32": "", "f32x3": "_x3", "f16x2": "_h2w"}[mode],
                      2.0 * S * T * M * Cn, launch)
    else:
        launch()
    return part


def sum_over_batch(x: torch.Tensor) -> torch.Tensor:
    _req(x, "sum_over_batch.x")
    B = x.shape[0]
    out = torch.empty(x.shape[1:], device=x.device, dtype=torch.float32)
    check(lib().bm_sum_over_batch(_p(x), _p(out), B, out.numel(), _stream()), "bm_sum_over_batch")
    return out


# ------------------------------------------------------------------------------------------------
def bn_finalize(stats, count: int, gamma, beta, running_mean, running_var, num_batches,
                momentum: float, eps: float):
    _req(stats, "bn_finalize.stats")
    channel_major = getattr(stats, "_bm_channel_major", False)
    (C, ntiles, _) = stats.shape if channel_major else (stats.shape[1], stats.shape[0], 2)
    mean, invstd, scale, shift = (torch.empty(C, device=stats.device, dtype=torch.float32)
                                  for _ in range(4))
    check((lib().bm_bn_finalize_cm if channel_major else lib().bm_bn_finalize)(_p(stats), ntiles, C, count, _p(_opt(gamma, "gamma")),
                               _p(_opt(beta, "beta")), _p(_opt(running_mean, "running_mean")),
                               _p(_opt(running_var, "running_var")),
                               _p(_opt(num_batches, "num_batches", torch.int64)), momentum, eps,
                               _p(mean), _p(invstd), _p(scale), _p(shift), _stream()),
          "bm_bn_finalize")
    return mean, invstd, scale, shift


def bn_eval_affine(gamma, beta, running_mean, running_var, eps: float):
    _req(running_mean, "running_mean")
    C = running_mean.numel()
    mean, invstd, scale, shift = (torch.empty(C, device=running_mean.device, dtype=torch.float32)
                                  for _ in range(4))
    check(lib().bm_bn_eval_affine(C, _p(_opt(gamma, "gamma")), _p(_opt(beta, "beta")),
                                  _p(running_mean), _p(_req(running_var, "running_var")), eps,
                                  _p(mean), _p(invstd), _p(scale), _p(shift), _stream()),
          "bm_bn_eval_affine")
    return mean, invstd, scale, shift


def affine_act_res(y, scale, shift, res, act: int, leak: float = 0.):
    _req(y, "affine_act_res.y")
    B, C, T = y.shape
    out = torch.empty_like(y)
    check(lib().bm_affine_act_res(_p(y), _p(_opt(scale, "scale")), _p(_opt(shift, "shift")),
                                  _p(_opt(res, "res")), _p(out), B, C, T, act, leak, *_slot_args(_amax_slot(out)),
                                  _stream()), "bm_affine_act_res")
    return out


def act_bn_bwd(dout, y, scale, shift, mean, invstd, bn_train: bool, act: int, leak: float = 0.,
               want_affine_grads: bool = False, want_dbias: bool = True):
    """Returns (dy, dgamma | None, dbeta | None, dbias | None)."""
    _req(dout, "act_bn_bwd.dout")
    _req(y, "act_bn_bwd.y")
    B, C, T = y.shape
    dy = torch.empty_like(y)
    dgamma = torch.empty(C, device=y.device, dtype=torch.float32) if want_affine_grads else None
    dbeta = torch.empty(C, device=y.device, dtype=torch.float32) if want_affine_grads else None
    dbias = torch.empty(C, device=y.device, dtype=torch.float32) if want_dbias else None
    nbytes = lib().bm_act_bn_bwd_workspace_bytes(B, C)
    ws = torch.empty(nbytes, device=y.device, dtype=torch.uint8)
    slot = _amax_slot(dy)
    check(lib().bm_act_bn_bwd(_p(dout), _p(y), _p(_opt(scale, "scale")), _p(_opt(shift, "shift")),
                              _p(_opt(mean, "mean")), _p(_opt(invstd, "invstd")), int(bn_train),
                              _p(dy), _p(dgamma), _p(dbeta), _p(dbias), _p(ws), nbytes, B, C, T, act,
                              leak, *_slot_args(slot), _p(_row_amax_out(dy, slot)), _stream()), "bm_act_bn_bwd")
    return dy, dgamma, dbeta, dbias


def channel_sum(x: torch.Tensor) -> torch.Tensor:
    _req(x, "channel_sum.x")
    B, C, T = x.shape
    out = torch.empty(C, device=x.device, dtype=torch.float32)
    nbytes = lib().bm_channel_sum_workspace_bytes(B, C)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    check(lib().bm_channel_sum(_p(x), C * T, _p(out), _p(ws), nbytes, B, C, T, _stream()),
          "bm_channel_sum")
    return out


def channel_stats(x: torch.Tensor) -> torch.Tensor:
    """[B, C, T] -> per-split (sum, sumsq) partials [nsplit, C, 2] for bn_finalize."""
    _req(x, "channel_stats.x")
    B, C, T = x.shape
    stats = torch.empty(lib().bm_channel_stats_splits(B), C, 2, device=x.device, dtype=torch.float32)
    check(lib().bm_channel_stats(_p(x), _p(stats), B, C, T, _stream()), "bm_channel_stats")
    return stats


def time_sums_t(x: torch.Tensor) -> torch.Tensor:
    """[B, C, T] -> [C, B]: out[c][b] = sum_t x[b][c][t]."""
    _req(x, "time_sums_t.x")
    B, C, T = x.shape
    out = torch.empty(C, B, device=x.device, dtype=torch.float32)
    check(lib().bm_time_sums_t(_p(x), _p(out), B, C, T, _stream()), "bm_time_sums_t")
    return out


def glu_fwd(u: torch.Tensor) -> torch.Tensor:
    _req(u, "glu_fwd.u")
    B, C2, T = u.shape
    out = torch.empty(B, C2 // 2, T, device=u.device, dtype=torch.float32)
    check(lib().bm_glu_fwd(_p(u), _p(out), B, C2 // 2, T, *_slot_args(_amax_slot(out)), _stream()), "bm_glu_fwd")
    return out


def glu_bwd(dout: torch.Tensor, u: torch.Tensor, want_dbias: bool = True):
    _req(dout, "glu_bwd.dout")
    _req(u, "glu_bwd.u")
    B, C2, T = u.shape
    H = C2 // 2
    du = torch.empty_like(u)
    dbias = torch.empty(C2, device=u.device, dtype=torch.float32) if want_dbias else None
    nbytes = lib().bm_glu_bwd_workspace_bytes(B, H)
    ws = torch.empty(nbytes, device=u.device, dtype=torch.uint8)
    slot = _amax_slot(du)
    check(lib().bm_glu_bwd(_p(dout), _p(u), _p(du), _p(dbias), _p(ws), nbytes, B, H, T,
                           *_slot_args(slot), _p(_row_amax_out(du, slot)), _stream()), "bm_glu_bwd")
    return du, dbias


# ------------------------------------------------------------------------------------------------
def fourier_emb(positions: torch.Tensor, D: int, margin: float = 0.2) -> torch.Tensor:
    _req(positions, "fourier_emb.positions")
    rows = positions.numel() // 2
    emb = torch.empty(*positions.shape[:-1], D, device=positions.device, dtype=torch.float32)
    check(lib().bm_fourier_emb(_p(positions), _p(emb), rows, D, margin, _stream()), "bm_fourier_emb")
    return emb


def masked_softmax(scores, positions, ban_center, ban_radius: float) -> torch.Tensor:
    _req(scores, "masked_softmax.scores")
    _req(positions, "masked_softmax.positions")
    U, O, C = scores.shape
    w = torch.empty_like(scores)
    check(lib().bm_masked_softmax(_p(scores), _p(positions), _p(_opt(ban_center, "ban_center")),
                                  float(ban_radius), _p(w), U, O, C, _stream()), "bm_masked_softmax")
    return w


def softmax_bwd(w, dw) -> torch.Tensor:
    _req(w, "softmax_bwd.w")
    _req(dw, "softmax_bwd.dw")
    ds = torch.empty_like(w)
    C = w.shape[-1]
    check(lib().bm_softmax_bwd(_p(w), _p(dw), _p(ds), w.numel() // C, C, _stream()),
          "bm_softmax_bwd")
    return ds


# ------------------------------------------------------------------------------------------------
def clip_inv_norms(cand: torch.Tensor, nonfinite_flag: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
    """1 / (1e-8 + ||cand[o]||) per candidate (bm/losses.py:91).  ONE pass over the candidates also yields what else
    a step needs from them: max |cand| (attached to the tensor like ``amax`` would, f16x2 mode) and -- with
    ``nonfinite_flag`` -- the reference's finiteness assert.  Cached on the tensor object together with its version
    counter: the Solver runs it when the batch arrives, ClipLoss finds the result."""
    cached = getattr(cand, "_bm_inv_norms", None)
    if cached is not None and cached[0] == cand._version and cached[1] == cand.data_ptr() and \
            (nonfinite_flag is None or cached[3]):
        return cached[2]
    _req(cand, "clip_inv_norms.cand")
    Bc = cand.shape[0]
    K = cand.numel() // max(Bc, 1)
    out = torch.empty(Bc, device=cand.device, dtype=torch.float32)
    have = getattr(cand, "_bm_amax", None)
    have = have is not None and have[0] == cand._version and have[1] == cand.data_ptr() and \
        (nonfinite_flag is None or have[3])
    slot = None if have else _amax_slot(cand)            # None outside f16x2 mode
    check(lib().bm_clip_cand_prep(_p(cand), Bc, K, _p(out), _p(slot),
                                  _p(_opt(nonfinite_flag, "nonfinite_flag", torch.int32)), _stream()),
          "bm_clip_cand_prep")
    if slot is not None:
        cand._bm_amax = (cand._version, cand.data_ptr(), slot, nonfinite_flag is not None)
    try:
        cand._bm_inv_norms = (cand._version, cand.data_ptr(), out, nonfinite_flag is not None)
    except Exception:
        pass
    return out


def flag_unless_all_set(mask: torch.Tensor, flag: torch.Tensor) -> None:
    """flag[0] |= 1 (device side, no sync) when the bool ``mask`` holds a False."""
    _req(mask, "flag_unless_all_set.mask", torch.bool)
    check(lib().bm_flag_unless_all_set(_p(mask), mask.numel(), _p(_req(flag, "flag", torch.int32)), _stream()),
          "bm_flag_unless_all_set")


def clip_ce(part, inv_norm, want_probs=False, want_grad=False, want_loss=False,
            target_offset: int = 0, col_valid: tp.Optional[torch.Tensor] = None):
    """part [nsplit][B][B'] -> (scores, probs|None, dscaled|None, loss|None).  ``col_valid`` ([B'] fp32, optional):
    candidates with 0 there are masked out of every row (score -inf, probability 0, gradient 0)."""
    _req(part, "clip_ce.part")
    nsplit, B, Bc = part.shape
    dev = part.device
    scores = torch.empty(B, Bc, device=dev, dtype=torch.float32)
    probs = torch.empty(B, Bc, device=dev, dtype=torch.float32) if want_probs else None
    dscaled = torch.empty(B, Bc, device=dev, dtype=torch.float32) if want_grad else None
    loss_row = torch.empty(B, device=dev, dtype=torch.float32) if want_loss else None
    loss = torch.empty((), device=dev, dtype=torch.float32) if want_loss else None
    check(lib().bm_clip_ce_masked(_p(part), nsplit, _p(_req(inv_norm, "inv_norm")), _p(_opt(col_valid, "col_valid")),
                                  _p(scores), _p(probs), _p(dscaled), _p(loss_row), _p(loss), B, Bc, target_offset,
                                  _stream()), "bm_clip_ce")
    return scores, probs, dscaled, loss


def clip_ce_cols(scores, inv_norm, dscaled, loss, target_offset: int = 0, w_row: float = 0.5, w_col: float = 0.5):
    """Column term of the symmetric objective on top of ``clip_ce``'s outputs: ``dscaled`` (row term) and ``loss``
    (row loss) are updated in place to the weighted sums; returns the per-target column losses [B]."""
    _req(scores, "clip_ce_cols.scores")
    B, Bc = scores.shape
    loss_col = torch.empty(B, device=scores.device, dtype=torch.float32)
    check(lib().bm_clip_ce_cols(_p(scores), _p(_req(inv_norm, "inv_norm")), _p(_opt(dscaled, "dscaled")),
                                _p(loss_col), _p(_opt(loss, "loss")), B, Bc, target_offset, w_row, w_col, _stream()),
          "bm_clip_ce_cols")
    return loss_col


def clip_cand_coef(dscaled, scores, inv_norm, alpha=None) -> torch.Tensor:
    _req(dscaled, "clip_cand_coef.dscaled")
    _req(scores, "clip_cand_coef.scores")
    B, Bc = dscaled.shape
    coef = torch.empty(Bc, device=dscaled.device, dtype=torch.float32)
    check(lib().bm_clip_cand_coef(_p(dscaled), _p(scores), _p(_req(inv_norm, "inv_norm")),
                                  _p(_opt(alpha, "alpha")), _p(coef), B, Bc, _stream()),
          "bm_clip_cand_coef")
    return coef


def row_axpy_sub(y: torch.Tensor, x: torch.Tensor, coef: torch.Tensor):
    """y[r] -= coef[r] * x[r] (in place), rows = first dimension."""
    _req(y, "row_axpy_sub.y")
    _req(x, "row_axpy_sub.x")
    rows = y.shape[0]
    check(lib().bm_row_axpy_sub(_p(y), _p(x), _p(_req(coef, "coef")), rows, y.numel() // max(rows, 1),
                                _stream()), "bm_row_axpy_sub")
    return _touched(y)


def center_scale(x: torch.Tensor, center: torch.Tensor, scale: torch.Tensor,
                 group: tp.Optional[torch.Tensor] = None, clip: bool = False, limit: float = 0.,
                 want_maxabs: bool = False, inplace: bool = False):
    """(x - center[group[b]]) / scale[group[b]] [clamped]; returns (out, maxabs [B] | None)."""
    _req(x, "center_scale.x")
    _req(center, "center_scale.center")
    _req(scale, "center_scale.scale")
    B, C, T = x.shape
    assert center.shape[-1] == C and scale.shape == center.shape
    out = _touched(x) if inplace else torch.empty_like(x)
    maxabs = torch.zeros(B, device=x.device, dtype=torch.float32) if want_maxabs else None
    check(lib().bm_center_scale(_p(x), _p(out), _p(_opt(group, "group", torch.int64)), _p(center),
                                _p(scale), B, C, T, int(clip), float(limit), _p(maxabs), _stream()),
          "bm_center_scale")
    return out, maxabs


def row_softmax(x: torch.Tensor) -> torch.Tensor:
    _req(x, "row_softmax.x")
    y = torch.empty_like(x)
    check(lib().bm_row_softmax(_p(x), _p(y), x.shape[0], x.shape[1], _stream()), "bm_row_softmax")
    return y


def rowwise_dot(a: torch.Tensor, b: torch.Tensor, scale=None) -> torch.Tensor:
    _req(a, "rowwise_dot.a")
    _req(b, "rowwise_dot.b")
    rows = a.shape[0]
    out = torch.empty(rows, device=a.device, dtype=torch.float32)
    check(lib().bm_rowwise_dot(_p(a), _p(b), _p(_opt(scale, "scale")), _p(out), rows,
                               a.numel() // max(rows, 1), _stream()), "bm_rowwise_dot")
    return out


def segment_sum_cols(p: torch.Tensor, order: torch.Tensor, seg: torch.Tensor) -> torch.Tensor:
    _req(p, "segment_sum_cols.p")
    rows, cols = p.shape
    V = seg.numel() - 1
    pv = torch.empty(rows, V, device=p.device, dtype=torch.float32)
    check(lib().bm_segment_sum_cols(_p(p), _p(_req(order, "order", torch.int32)),
                                    _p(_req(seg, "seg", torch.int32)), _p(pv), rows, cols, V, _stream()),
          "bm_segment_sum_cols")
    return pv


def topk_rows(x: torch.Tensor, k: int, col_labels=None, row_labels=None):
    """x [N, V] -> (idx [N, k] int32, values [N, k], hits [N] int32 | None)."""
    _req(x, "topk_rows.x")
    N, V = x.shape
    idx = torch.empty(N, k, device=x.device, dtype=torch.int32)
    val = torch.empty(N, k, device=x.device, dtype=torch.float32)
    hits = torch.empty(N, device=x.device, dtype=torch.int32) if col_labels is not None else None
    check(lib().bm_topk_rows(_p(x), N, V, k, _p(idx), _p(val),
                             _p(_opt(col_labels, "col_labels", torch.int64)),
                             _p(_opt(row_labels, "row_labels", torch.int64)), _p(hits), _stream()),
          "bm_topk_rows")
    return idx, val, hits


def adam_step(param, grad, exp_avg, exp_avg_sq, step: int, lr: float, beta1: float, beta2: float,
              eps: float, grad_scale: float = 1.0):
    for t, n in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _req(t, f"adam_step.{n}")
    check(lib().bm_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), step,
                             lr, beta1, beta2, eps, grad_scale, _stream()), "bm_adam_step")
    weights_changed()
