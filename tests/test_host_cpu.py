"""CPU-only checks of the host side: the C-ABI library loads and exports every declared symbol,
the modules keep the reference's state_dict contract, the product refuses to run without the GPU
(no fallback), and the data-parallel exchange is correct at world_size 2 on gloo."""
import os
import re
import subprocess
import sys
from pathlib import Path

import pytest
import torch

from helpers import Golden, MODEL_FIXTURES, OFF_PATH_FIXTURES
from oracle import bm_oracle as O

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from brainmagick_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 30
    handle = _lib.lib()
    for name in protos:
        assert hasattr(handle, name), name
    assert handle.bm_version() >= 100
    # geometry helpers are pure host functions: callable without a GPU
    assert handle.bm_conv_mpad(320) == 320 and handle.bm_conv_mpad(270) == 288
    assert handle.bm_conv_stats_tiles(256, 360) == 768
    assert handle.bm_bwd_nsplit(256) == 8


def test_header_cites_reference_for_every_group():
    text = (ROOT / "include" / "bm_hip.h").read_text()
    assert len(re.findall(r"(?:bm/)?[\w/]+\.py:\d+", text)) >= 10


@pytest.mark.parametrize("name", MODEL_FIXTURES + OFF_PATH_FIXTURES)
def test_state_dict_contract(name):
    """Same keys, shapes and dtypes as the reference's state_dict; strict load works; same-seed
    construction reproduces the reference initialisation bit for bit."""
    from brainmagick_amd.models import SimpleConv
    g = Golden(name)
    meta = g.meta
    sd0 = g.group("sd0")
    seed = 2036 + sum(map(ord, name))            # tests/golden/make_golden.py
    torch.manual_seed(seed)
    model = SimpleConv(in_channels={"meg": meta["C"], **meta.get("extra_inputs", {})}, out_channels=meta["F"],
                       hidden={"meg": meta["hidden"], **meta.get("extra_hidden", {})}, n_subjects=meta["S"],
                       **meta["cfg"])
    sd = model.state_dict()
    assert list(sd.keys()) == list(sd0.keys())
    bn_prefixes = {k.rsplit(".", 1)[0] for k in sd if k.endswith(".running_mean")}
    for k in sd:
        assert sd[k].shape == sd0[k].shape and sd[k].dtype == sd0[k].dtype, k
        if k.rsplit(".", 1)[0] not in bn_prefixes:
            assert torch.equal(sd[k], sd0[k]), k   # BN tensors were randomised after construction
    model.load_state_dict(sd0, strict=True)


def test_unsupported_options_raise():
    from brainmagick_amd.models import SimpleConv
    base = dict(in_channels={"meg": 8}, out_channels=4, hidden={"meg": 8})
    for kw in (dict(n_fft=16),):
        with pytest.raises(NotImplementedError):
            SimpleConv(**{**base, "in_channels": {"meg": 8}}, **kw)
    with pytest.raises(ValueError):
        SimpleConv(in_channels={"meg": 8}, out_channels=4, hidden={"eeg": 8})
    with pytest.raises(ValueError):
        SimpleConv(in_channels={"eeg": 8}, out_channels=4, hidden={"eeg": 8})
    with pytest.raises(AssertionError):          # two stacks need a head to merge them (reference assert)
        SimpleConv(in_channels={"meg": 8, "aux": 2}, out_channels=4, hidden={"meg": 8, "aux": 4})


def test_no_cpu_fallback():
    from brainmagick_amd.models import SimpleConv
    from brainmagick_amd.losses import ClipLoss
    from brainmagick_amd.optim import FlatAdam
    from brainmagick_amd import synthetic
    model = SimpleConv(in_channels={"meg": 8}, out_channels=4, hidden={"meg": 8}, subject_dim=0)
    sb = synthetic.make_batch(2, 8, 16, 4, 3)
    with pytest.raises(RuntimeError):
        model({"meg": sb.meg}, sb)
    with pytest.raises(RuntimeError):
        ClipLoss()(torch.randn(2, 4, 16), torch.randn(2, 4, 16), torch.ones(2, 1, 16, dtype=torch.bool))
    with pytest.raises(RuntimeError):
        FlatAdam(model.parameters())


def test_clip_loss_asserts_like_reference():
    from brainmagick_amd.losses import ClipLoss
    loss = ClipLoss()
    with pytest.raises(AssertionError):
        loss(torch.randn(3, 2, 4), torch.randn(2, 2, 4), torch.ones(3, 1, 4, dtype=torch.bool))
    with pytest.raises(AssertionError):
        loss(torch.randn(2, 2, 4), torch.randn(2, 2, 4), torch.zeros(2, 1, 4, dtype=torch.bool))
    assert ClipLoss(linear=16).linear is None        # accepted, never applied (losses.py:35,82)


def test_product_never_imports_oracle():
    for path in (ROOT / "brainmagick_amd").rglob("*.py"):
        text = path.read_text()
        assert "import oracle" not in text and "from oracle" not in text, path


def test_synthetic_batches_are_seeded_and_shaped():
    from brainmagick_amd import synthetic
    a = synthetic.make_config_batch("cfg2", batch=4)
    b = synthetic.make_config_batch("cfg2", batch=4)
    assert torch.equal(a.meg, b.meg) and a.meg.shape == (4, 208, 360)
    assert a.features.shape == (4, 120, 360) and a.meg.abs().max() <= 20
    m = synthetic.make_config_batch("cfg5", batch=16)
    pos = m.positions()
    eeg = [i for i, r in enumerate(m._recordings) if len(r.layout) == 128]
    assert eeg, "mixed batch should contain EEG segments"
    assert (pos[eeg[0], 128:] == synthetic.INVALID).all() and (m.meg[eeg[0], 128:] == 0).all()
    assert (m.subject_index[eeg] < 19).all()


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from brainmagick_amd import distrib
from oracle import bm_oracle as O

class CpuFlatAdam:
    """Test double with FlatAdam's interface, arithmetic = the oracle's adam_step."""
    def __init__(self, n, pad_to):
        self.padded = (n + pad_to - 1) // pad_to * pad_to
        g = torch.Generator().manual_seed(0)
        self.flat_param = torch.randn(self.padded, generator=g)
        self.flat_grad = torch.zeros(self.padded)
        self.m = torch.zeros(self.padded); self.v = torch.zeros(self.padded); self.t = 0
    def step(self, shard=None, grad_scale=1.0):
        self.t += 1
        lo, hi = shard if shard else (0, self.padded)
        O.adam_step(self.flat_param[lo:hi], self.flat_grad[lo:hi] * grad_scale, self.m[lo:hi],
                    self.v[lo:hi], self.t)

distrib.init("gloo")
r, w = distrib.rank(), distrib.world_size()
assert w == 2
opt = CpuFlatAdam(1001, pad_to=w * 4)
ref = CpuFlatAdam(1001, pad_to=w * 4)
for it in range(3):
    grads = [torch.randn(opt.padded, generator=torch.Generator().manual_seed(100 * it + k)) for k in range(w)]
    opt.flat_grad.copy_(grads[r])
    distrib.sharded_step(opt)
    ref.flat_grad.copy_(O.sync_gradients_reference([[g] for g in grads])[0])
    ref.step()
assert torch.allclose(opt.flat_param, ref.flat_param, rtol=1e-6, atol=1e-7), (opt.flat_param - ref.flat_param).abs().max()
# candidate all-gather: rank-ordered, own block at rank*B
gather = distrib.CandidateGather()
cand = torch.full((3, 2, 5), float(r))
gather.start(cand)
out, off, valid = gather.wait()
assert out.shape == (6, 2, 5) and off == 3 * r and valid is None
assert (out[off:off + 3] == r).all() and (out[3 * (1 - r):3 * (1 - r) + 3] == 1 - r).all()
# ragged blocks (per-rank rejection): rank r brings 3 - r of nominally 3 candidates; padding rows are flagged invalid
gather.start(torch.full((3 - r, 2, 5), float(r + 1)), block_rows=3)
out, off, valid = gather.wait()
assert out.shape == (6, 2, 5) and off == 3 * r
assert valid.tolist() == [1., 1., 1., 1., 1., 0.], valid
assert (out[:3] == 1).all() and (out[3:5] == 2).all() and (out[5] == 0).all()
m = distrib.average_metrics({"loss": float(r + 1)}, count=r + 1)
assert abs(m["loss"] - (1 * 1 + 2 * 2) / 3) < 1e-9
# BN buffers are averaged
bn = torch.nn.BatchNorm1d(4); bn.running_mean.fill_(float(r))
distrib.sync_buffers(bn)
assert torch.allclose(bn.running_mean, torch.full((4,), 0.5))
# flashy.sync_model's buffer averaging as ONE flat all-reduce (buffers become views of the bucket)
bn2 = torch.nn.BatchNorm1d(3); bn2.running_var.fill_(float(2 * r + 1))
bucket = distrib.BufferBucket([bn2])
bucket.average()
assert torch.allclose(bn2.running_var, torch.full((3,), 2.0)) and bn2.running_var.data_ptr() >= bucket.flat.data_ptr()
# sharded optimizer moments: every rank owns shard r, a checkpoint needs all of them
mom = torch.zeros(8); lo, hi = distrib.shard_bounds(8, w, r); mom[lo:hi] = r + 1.0
distrib.all_gather_shards(mom)
assert torch.equal(mom, torch.tensor([1.] * 4 + [2.] * 4)), mom
assert abs(distrib.max_over_ranks(float(r)) - 1.0) < 1e-9
distrib.check_equal_over_ranks(7, "same everywhere")
try:
    distrib.check_equal_over_ranks(7 + r, "differs")
    raise SystemExit("check_equal_over_ranks did not raise")
except RuntimeError:
    pass
# learnable candidates: autograd-aware all-gather, gradient reduce-scattered back to the owner
x = torch.full((2, 3), float(r + 1), requires_grad=True)
gathered, off = distrib.gather_learnable_candidates(x)
assert off == 2 * r and gathered.shape == (4, 3)
wts = torch.arange(12.).view(4, 3) * (r + 1)
(gathered * wts).sum().backward()
expect = sum(torch.arange(12.).view(4, 3)[2 * r:2 * r + 2] * (k + 1) for k in range(w))
assert torch.equal(x.grad, expect), (x.grad, expect)
dist.barrier()
print("WORKER_OK", r)
'''


def test_data_parallel_exchange_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), str(ROOT)], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"WORKER_OK {r}" in out, out


_WIDE = ("gemm_nt_h2w.hip", "conv_nn_h2w.hip", "conv_nn_h2d.hip")
_asm_cache = {}


def _wide_kernel_asm():
    """gfx950 assembly of the wide-tile kernels (compiled once per session, the two files in parallel)."""
    import concurrent.futures
    import shutil
    import subprocess
    import tempfile
    if _asm_cache:
        return _asm_cache
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("hipcc not available")
    csrc = ROOT / "brainmagick_amd" / "csrc"
    tmp = Path(tempfile.mkdtemp(prefix="bm_asm_"))

    def one(name):
        out = tmp / (name + ".s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        f"-I{csrc}", "-o", str(out), str(csrc / name)], check=True, capture_output=True)
        return name, out.read_text()
    with concurrent.futures.ThreadPoolExecutor(4) as pool:
        _asm_cache.update(dict(pool.map(one, _WIDE)))
    return _asm_cache


def test_wide_kernels_do_not_spill():
    """The wide kernels keep 240 accumulators + operand fragments + staging sets in the 512-register file; a
    spill would put scratch traffic (and, for the asm-issued loads, silent corruption) into the main loop.  The LDS-DMA
    conv (hand-issued loads) must have none at all; gemm_nt_h2w keeps a few scratch bytes for its once-per-launch tail
    stage and the residual epilogue of the production conv a few for its 48 in-flight residual values (compiler-visible
    loads: slow-path only, never a hazard) -- there the MAIN LOOP (first to last MFMA) must be free of scratch accesses."""
    import re
    for name, text in _wide_kernel_asm().items():
        spills = [int(l.split(":")[1]) for l in text.splitlines() if ".vgpr_spill_count" in l]
        scratch = [int(l.split(":")[1]) for l in text.splitlines() if ".private_segment_fixed_size" in l]
        assert spills and scratch, name
        if name == "conv_nn_h2d.hip":
            assert all(s == 0 for s in spills) and all(s == 0 for s in scratch), (name, spills, scratch)
            continue
        assert all(s <= 64 for s in scratch), (name, scratch)
        if name == "conv_nn_h2w.hip":
            for kname, body in re.findall(r"^(_Z\d+conv_nn_h2w_kernel\w+):(.*?)^\.Lfunc_end", text, flags=re.M | re.S):
                lines = body.splitlines()
                mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
                inside = [l for l in lines[mf[0]:mf[-1] + 1] if "scratch_" in l]
                assert not inside, (kname, inside[:3])


def test_hand_issued_loads_are_not_touched_before_their_wait():
    """conv_nn_h2d (the LDS-DMA main loop kept for A/B runs) issues the input-window loads through inline asm with
    hand-counted waits (a compiler-visible load next to the LDS-DMA weight copies would drain the DMA queue at every
    use).  hipcc may copy or re-use an asm load's destination register while the load is still in flight;
    scripts/audit_asm_loads.py checks on the generated ISA that it does not.  (gemm_nt_h2w and, since round 6, the
    production conv_nn_h2w use compiler-visible loads only: there is no LDS-DMA next to them any more.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("audit_asm_loads", ROOT / "scripts" / "audit_asm_loads.py")
    audit = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(audit)
    asm = _wide_kernel_asm()
    for name in ("conv_nn_h2d.hip",):
        violations = audit.audit_text(asm[name])
        assert not violations, (name, violations[:5])
    for name in ("gemm_nt_h2w.hip", "conv_nn_h2w.hip"):
        in_asm, hidden = False, []
        for line in asm[name].splitlines():
            if "ASMSTART" in line:
                in_asm = True
            elif "ASMEND" in line:
                in_asm = False
            elif in_asm and ("buffer_load" in line or "global_load" in line):
                hidden.append(line.strip())
        assert not hidden, (name, hidden[:3])


def test_wide_conv_accumulates_in_place():
    """The production conv keeps its MW x 3 accumulator blocks in the same AGPRs for the whole tile: every MFMA of
    the main loop writes the block it reads.  (A peeled tail chunk behind the chunk-pair loop once made hipcc rotate
    the blocks through copies -- 200 v_accvgpr moves and 88 scratch accesses per loop trip in the 1x1 kernels.)"""
    import re
    asm = _wide_kernel_asm()["conv_nn_h2w.hip"]
    kernels = re.findall(r"^(_Z\d+conv_nn_h2w_kernel\w+):.*?^\.Lfunc_end", asm, flags=re.M | re.S)
    bodies = re.findall(r"^_Z\d+conv_nn_h2w_kernel\w+:(.*?)^\.Lfunc_end", asm, flags=re.M | re.S)
    assert len(kernels) == 9 and len(bodies) == 9, kernels          # (1x1, 3 taps, 3 taps + residual epilogue) x MW in {5, 4, 2}
    for name, body in zip(kernels, bodies):
        mfmas = re.findall(r"v_mfma_f32_32x32x16_f16 (a\[\d+:\d+\]), [^,]+, [^,]+, (\S+)", body)
        assert len(mfmas) >= 36, (name, len(mfmas))      # 1x1, MW = 2: two stages of 18
        moved = [m for m in mfmas if m[0] != m[1]]
        assert not moved, (name, moved[:3])


def test_deep_mel_known_answer_of_the_reference():
    """The one test the reference holds for this path (bm/test_model.py:53-68): DeepMel built from the same
    arguments has `n_hidden_layers` blocks; the output shape half of it runs on the GPU
    (tests/test_model_gpu.py::test_deep_mel_shape_like_reference)."""
    from brainmagick_amd.models import DeepMel
    model = DeepMel(8, 3, 5, 2, kernel=3, stride=1, dilation_growth=2, dilation_period=5, batch_norm=True,
                    activation_on_last=False, skip=True, glu_context=1, glu=2)
    assert len(model.sequence) == 5


def test_layout_device_cache_hits_on_the_second_call():
    """the cache key was clobbered by the loop variable, so every step re-uploaded the layouts."""
    from brainmagick_amd import synthetic
    from brainmagick_amd.models.common import PositionGetter
    batch = synthetic.make_batch(6, 10, 8, 4, 3, n_layouts=3)
    getter = PositionGetter()
    pos1, idx1 = getter.get_unique_layouts(batch, 10, "cpu")
    pos2, idx2 = getter.get_unique_layouts(batch, 10, "cpu")
    assert pos1 is pos2 and idx1 is idx2
    assert len(getter._device_cache) == 1 and isinstance(next(iter(getter._device_cache)), tuple)
    want = torch.stack([batch._recordings[i].layout for i in range(6)])
    assert torch.equal(pos1[idx1], want)


def test_scale_reject_subset_follows_the_reference_getitem():
    """bm/dataset.py:242-257: batch[keep] subsets every list field and tolerates empty lists."""
    import dataclasses
    from brainmagick_amd import synthetic
    from brainmagick_amd.norm import _subset
    batch = synthetic.make_batch(5, 4, 8, 3, 2, n_layouts=2)
    batch._event_lists = [[f"ev{i}"] for i in range(5)]
    keep = torch.tensor([True, False, True, True, False])
    sub = _subset(batch, keep)
    assert len(sub) == 3 and sub._event_lists == [["ev0"], ["ev2"], ["ev3"]]
    assert [r.recording_index for r in sub._recordings] == [batch._recordings[i].recording_index for i in (0, 2, 3)]
    assert torch.equal(sub.meg, batch.meg[keep]) and torch.equal(sub.subject_index, batch.subject_index[keep])

    @dataclasses.dataclass
    class Plain:          # a batch type without __getitem__ and with empty optional lists
        meg: torch.Tensor
        features_mask: torch.Tensor
        _recordings: list = dataclasses.field(default_factory=list)
    plain = _subset(Plain(batch.meg, batch.features_mask), keep)
    assert plain._recordings == [] and plain.meg.shape[0] == 3


def test_pack_plan_refresh_and_eviction_logic(monkeypatch):
    """hip_ops._PackPlan (persistent packed parameters, one batched launch per parameter update) against a stub
    library: one refresh per update however many parameters ask, torch-side edits are seen through the version
    counter, parameters nobody asks for any more leave the plan."""
    from brainmagick_amd import hip_ops as H

    class StubLib:
        def __init__(self):
            self.batches = []

        def bm_packed_weight_bytes_h2(self, G, M, Cin, KS):
            return 64

        def bm_pack_h2_job_bytes(self):
            return 16

        def bm_pack_h2_job_fill(self, job, src, dst, *rest):
            return 10

        def bm_pack_weights_h2_batch(self, table, njobs, total_blocks, max_nk, stream):
            assert max_nk == 9
            self.batches.append((njobs, total_blocks))
            return 0

    stub = StubLib()
    monkeypatch.setattr(H, "lib", lambda: stub)
    monkeypatch.setattr(H, "_stream", lambda: None)
    monkeypatch.setattr(H, "_weights_epoch", 0)
    plan = H._PackPlan(torch.device("cpu"))
    params = [torch.nn.Parameter(torch.randn(4, 3, 3)) for _ in range(3)]
    geom = (1, 4, 3, 3, 0, 9, 3, 1, 0)
    packed = [plan.get(p, geom) for p in params]
    n_registration = len(stub.batches)                    # registration refreshes (one per new parameter)
    assert stub.batches[-1] == (3, 30)
    assert [plan.get(p, geom) is q for p, q in zip(params, packed)] == [True] * 3
    assert len(stub.batches) == n_registration            # nothing changed: served from the plan
    H.weights_changed()
    for p in params:
        plan.get(p, geom)
    assert len(stub.batches) == n_registration + 1        # one launch for all three
    with torch.no_grad():
        params[1].mul_(2.0)                               # torch-side edit: version counter
    plan.get(params[0], geom)
    assert len(stub.batches) == n_registration + 1
    plan.get(params[1], geom)
    assert len(stub.batches) == n_registration + 2
    # a parameter that is not asked for during KEEP + 1 updates leaves the plan
    for _ in range(H._PackPlan.KEEP + 2):
        H.weights_changed()
        plan.get(params[0], geom)
        plan.get(params[1], geom)
    assert stub.batches[-1][0] == 2 and len(plan.entries) == 2


def test_loopback_communicator_runs_the_data_parallel_exchange_in_process():
    """tests/loopback.py (N replicas as threads of one process, the stand-in for ranks on the 1-GPU test box) against
    the same expectations as the 2-process gloo test: sharded step == mean-gradient Adam, rank-ordered candidate
    gather (also with ragged, padded blocks), buffer averaging, moment gather -- here at world 2, 3, 4 and 8, with and
    without optimizer sharding."""
    from loopback import run_replicas
    from brainmagick_amd import distrib

    class CpuFlatAdam:
        def __init__(self, n, pad_to):
            self.padded = (n + pad_to - 1) // pad_to * pad_to
            g = torch.Generator().manual_seed(0)
            self.flat_param = torch.randn(self.padded, generator=g)
            self.flat_grad = torch.zeros(self.padded)
            self.m = torch.zeros(self.padded)
            self.v = torch.zeros(self.padded)
            self.t = 0

        def step(self, shard=None, grad_scale=1.0):
            self.t += 1
            lo, hi = shard if shard else (0, self.padded)
            O.adam_step(self.flat_param[lo:hi], self.flat_grad[lo:hi] * grad_scale, self.m[lo:hi], self.v[lo:hi],
                        self.t)

    for world in (2, 3, 4, 8):
        for shard in (True, False):
            def body(r):
                assert distrib.is_distributed() and distrib.world_size() == world and distrib.rank() == r
                opt = CpuFlatAdam(1001, pad_to=world * 4)
                for it in range(3):
                    g = torch.Generator().manual_seed(100 * it + r)
                    opt.flat_grad.copy_(torch.randn(opt.padded, generator=g))
                    distrib.sharded_step(opt, shard=shard)
                gather = distrib.CandidateGather()
                gather.start(torch.full((3, 2, 5), float(r)))
                out, off, valid = gather.wait()
                assert off == 3 * r and valid is None
                assert all(bool((out[3 * k:3 * k + 3] == k).all()) for k in range(world))
                # ragged blocks: rank r brings 1 + r % 3 of nominally 3 candidates, the rest of its block is padding
                gather.start(torch.full((1 + r % 3, 2, 5), float(r + 1)), block_rows=3)
                out, off, valid = gather.wait()
                assert off == 3 * r and out.shape[0] == 3 * world
                for k in range(world):
                    n = 1 + k % 3
                    assert valid[3 * k:3 * k + 3].tolist() == [1.] * n + [0.] * (3 - n)
                    assert bool((out[3 * k:3 * k + n] == k + 1).all()) and bool((out[3 * k + n:3 * k + 3] == 0).all())
                bn = torch.nn.BatchNorm1d(4)
                bn.running_mean.fill_(float(r))
                bucket = distrib.BufferBucket([bn])
                bucket.average()
                assert torch.allclose(bn.running_mean, torch.full((4,), (world - 1) / 2))
                distrib.all_gather_shards(opt.m)
                assert abs(distrib.max_over_ranks(float(r)) - (world - 1)) < 1e-9
                return opt.flat_param.clone(), opt.m.clone()

            res = run_replicas(world, body)
            ref = CpuFlatAdam(1001, pad_to=world * 4)
            for it in range(3):
                grads = [torch.randn(ref.padded, generator=torch.Generator().manual_seed(100 * it + k))
                         for k in range(world)]
                ref.flat_grad.copy_(O.sync_gradients_reference([[g] for g in grads])[0])
                ref.step()
            for p, m in res:
                assert torch.allclose(p, ref.flat_param, rtol=1e-6, atol=1e-7)
                assert torch.allclose(m, ref.m, rtol=1e-5, atol=1e-8)      # gathered moments are complete
            assert torch.equal(res[0][0], res[-1][0])                      # replicas stay bit-identical
    assert distrib.comm() is None and not distrib.is_distributed()


def test_shutdown_forgets_the_rendezvous_store():
    """a second init() in the same process (another MASTER_PORT / WORLD_SIZE) must not reuse the first
    rendezvous store."""
    from brainmagick_amd import distrib
    distrib._store_cache = object()
    distrib.shutdown()
    assert distrib._store_cache is None and distrib.comm() is None


def test_bench_self_launch_command():
    """`python bench.py --gpus N` outside torchrun re-launches itself as the driver would launch it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd, env = bench.self_launch_command(8, ["--gpus", "8", "--steps", "5"], port=29777)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29777"
    assert cmd[-5] == str(ROOT / "bench.py") and cmd[-4:] == ["--gpus", "8", "--steps", "5"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "RANK" not in env
    cmd2, _ = bench.self_launch_command(2, [])
    assert 1024 < int(cmd2[cmd2.index("--master-port") + 1]) < 65536


TORCHRUN_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from brainmagick_amd import distrib
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"       # what torch.distributed.run exports
# the unique-id exchange of the RCCL communicator (distrib._RcclComm) through the launcher's store
store = distrib._rendezvous_store(rank, world)
assert distrib._all_ranks_ok(store, "t/loaded", rank, world, True)
assert not distrib._all_ranks_ok(store, "t/one_failed", rank, world, rank != 1)    # agreed by EVERY rank
if rank == 0:
    store.set("t/id", bytes(range(128)))
assert bytes(store.get("t/id")) == bytes(range(128))
# fallback communicator on the SAME store (no second rendezvous on MASTER_PORT)
distrib._comm = distrib._TorchComm("gloo", store=store)
assert distrib.is_distributed() and distrib.world_size() == world and distrib.rank() == rank
t = torch.tensor([float(rank + 1)])
distrib.comm().all_reduce(t)
assert float(t) == 3.0
distrib.barrier()
distrib.shutdown()
print("TORCHRUN_WORKER_OK", rank, flush=True)
'''


def test_rendezvous_under_torch_distributed_run(tmp_path):
    """The launch the driver uses (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P`): the agent's TCP store carries the communicator's unique id and the agreement on
    the communicator kind, and the torch.distributed fallback joins through the same store."""
    import socket
    script = tmp_path / "torchrun_worker.py"
    script.write_text(TORCHRUN_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), str(ROOT)],
                         env=env, capture_output=True, text=True, timeout=240)
    # (the two ranks share stdout: their lines may interleave)
    assert out.returncode == 0 and out.stdout.count("TORCHRUN_WORKER_OK") == 2, out.stdout + out.stderr


def test_adopts_a_process_group_somebody_else_initialised(tmp_path):
    """a maintainer who keeps `flashy.distrib.init()` (= torch.distributed.init_process_group,
    bm/train.py:139) and only swaps `sharded_step` in must still get the gradient exchange."""
    worker = tmp_path / "adopt_worker.py"
    worker.write_text(r'''
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
dist.init_process_group("gloo")                      # NOT brainmagick_amd.distrib.init()
from brainmagick_amd import distrib
assert distrib.comm() is None
assert distrib.is_distributed() and distrib.world_size() == 2 and distrib.rank() == dist.get_rank()
assert distrib.comm_kind() == "torch.distributed/gloo"
class Opt:
    def __init__(self):
        self.flat_param = torch.zeros(8); self.flat_grad = torch.full((8,), float(dist.get_rank() + 1)); self.padded = 8
        self.calls = []
    def step(self, shard=None, grad_scale=1.0):
        self.calls.append((shard, grad_scale))
        lo, hi = shard
        self.flat_param[lo:hi] -= self.flat_grad[lo:hi] * grad_scale
opt = Opt()
distrib.sharded_step(opt)
assert opt.calls == [((4 * dist.get_rank(), 4 * dist.get_rank() + 4), 0.5)]
assert torch.equal(opt.flat_param, torch.full((8,), -1.5)), opt.flat_param     # mean of 1 and 2, all shards gathered
print("ADOPT_OK", dist.get_rank(), flush=True)
''')
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        procs.append(subprocess.Popen([sys.executable, str(worker), str(ROOT)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"ADOPT_OK {r}" in out, out


def test_f16x2_split_bound_on_the_cpu():
    """The arithmetic of the f16x2 contractions (power-of-two scale, two f16 planes, three products)
    emulated with numpy: norm-wise fp32-class always; with ONE scale per tensor a channel d decades below the maximum
    follows the documented 2^-37 * 10^d bound; with per-row scales every row is fp32-class.  (The GPU kernels are
    held to the same numbers by tests/test_f16x2_gpu.py::test_per_channel_spread_*.)"""
    import numpy as np

    def scale_from_amax(a):
        return 2.0 ** (14 - np.floor(np.log2(a)))

    def split(x, s):
        xs = (x * s).astype(np.float32)
        hi = xs.astype(np.float16)
        lo = (xs - hi.astype(np.float32)).astype(np.float16)
        return hi.astype(np.float64), lo.astype(np.float64)

    rng = np.random.default_rng(0)
    M, C, K = 48, 40, 2048
    for decades in (0, 4, 6, 8):
        sd = np.logspace(0, -decades, M)
        dy = (rng.standard_normal((M, K)) * sd[:, None]).astype(np.float32)
        x = rng.standard_normal((C, K)).astype(np.float32)
        ref = dy.astype(np.float64) @ x.astype(np.float64).T
        sx = scale_from_amax(np.abs(x).max())
        xh, xl = split(x, sx)
        # one scale per tensor
        sa = scale_from_amax(np.abs(dy).max())
        ah, al = split(dy, sa)
        got = (ah @ xh.T + ah @ xl.T + al @ xh.T) / (sa * sx)
        row = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 5e-7
        assert (row <= np.maximum(2e-6, 2.0 ** -37 / sd)).all(), (decades, (row * sd).max())
        if decades == 8:
            assert row.max() > 1e-5          # the bound is not vacuous: far-down rows do lose accuracy
        # one scale per row
        sar = np.array([scale_from_amax(np.abs(r).max()) for r in dy])
        ah, al = split(dy, sar[:, None])
        got = (ah @ xh.T + ah @ xl.T + al @ xh.T) / (sar[:, None] * sx)
        row = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
        assert row.max() < 5e-7, (decades, row.max())


def test_clip_candidate_blocks_cover_the_set_below_the_addressing_limit():
    """ClipLoss walks candidate sets beyond 1 GB in row blocks (2 048 wav2vec2-sized candidates of 8 ranks = 3 GB):
    the blocks tile the rows exactly, each slab stays below the kernels' 32-bit / staged-window limit, and the
    per-rank sizes of the paper's configurations need a single block."""
    from brainmagick_amd import functional as BF
    for Bc, K in [(256, 120 * 360), (256, 1024 * 360), (2048, 120 * 360)]:
        assert BF._candidate_blocks(Bc, K) == [(0, Bc)]
    for Bc, K in [(2048, 1024 * 360), (4096, 1024 * 360), (1000, 1024 * 361), (3, 200_000_000)]:
        blocks = BF._candidate_blocks(Bc, K)
        assert len(blocks) > 1 and blocks[0][0] == 0
        assert sum(n for _, n in blocks) == Bc
        assert all(r0 == sum(n for _, n in blocks[:i]) for i, (r0, _) in enumerate(blocks))
        assert all(n >= 1 and n * K * 4 < 0x40000000 or n == 1 for _, n in blocks)


def test_c_level_stdout_is_parked_on_stderr_while_a_communicator_comes_up():
    """librccl prints a version banner with printf when the first communicator is created; bench.py's contract is
    ONE JSON line on stdout.  `distrib._c_stdout_to_stderr` (wrapped around `bm_comm_init`) points fd 1 at stderr
    for the duration, C buffers flushed on both sides."""
    import subprocess
    code = (
        "import ctypes, sys\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "from brainmagick_amd import distrib\n"
        "libc = ctypes.CDLL(None)\n"
        "print('before')\n"
        "with distrib._c_stdout_to_stderr():\n"
        "    libc.printf(b'BANNER\\n')\n"
        "libc.printf(b'after-c\\n')\n"
        "libc.fflush(None)\n"
        "print('after')\n")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert p.stdout.split() == ["before", "after-c", "after"], p.stdout
    assert "BANNER" in p.stderr


def test_clip_loss_trim_keeps_the_tensors_when_nothing_is_cropped():
    """`ClipLoss.trim_samples` (bm/losses.py:50-75) slices both tensors; without tmin / tmax the slice is the
    identity, and the tensors themselves must go on -- a view is a new object without the maxima / candidate norms
    their producers attached (three stand-alone passes per step otherwise).  With a crop the result equals the
    reference's slice."""
    from types import SimpleNamespace
    from brainmagick_amd.losses import ClipLoss
    est, cand = torch.randn(3, 4, 50), torch.randn(5, 4, 50)
    loss = ClipLoss()
    a, b = loss.trim_samples(est, cand)
    assert a is est and b is cand
    loss = ClipLoss(tmin=0.1, tmax=0.3, dset_args=SimpleNamespace(tmin=-0.1, sample_rate=100))
    a, b = loss.trim_samples(est, cand)
    assert a.shape[-1] == 20 and torch.equal(a, est[..., 20:40]) and torch.equal(b, cand[..., 20:40])


def test_score_kernel_shape_rules():
    """Host-side rules of `bm_clip_scores_h2`: which shapes it takes, and its split counts (no GPU needed)."""
    from brainmagick_amd import _lib
    L = _lib.lib()
    T = 360
    assert L.bm_clip_scores_h2_covers(256, 256, 120 * T) and L.bm_clip_scores_h2_covers(256, 256, 1024 * T)
    assert L.bm_clip_scores_h2_covers(256, 2048, 120 * T)
    assert not L.bm_clip_scores_h2_covers(256, 2048, 1024 * T)        # 3 GB of candidates: walked in row blocks
    assert not L.bm_clip_scores_h2_covers(8, 8, 120 * T)              # a tile would be almost all padding
    assert not L.bm_clip_scores_h2_covers(256, 254, 120 * T)          # 16-byte partial stores need Bc % 4 == 0
    # short K: 128-estimate tiles, two per split -> 128 splits fill 256 CUs; long K: one 256 x 256 tile per split
    assert L.bm_clip_scores_h2_suggest_splits(256, 256, 120 * T) == 128
    assert L.bm_clip_scores_h2_suggest_splits(256, 256, 1024 * T) == 256
    assert L.bm_clip_scores_h2_suggest_splits(256, 2048, 120 * T) == 32
    assert L.bm_conv_h2_stats_tiles(256, 360) == 256 * 2 * 2


def test_header_prototypes_match_the_definitions():
    """`include/bm_hip.h` is what `_lib.py` parses into ctypes prototypes, but the kernels' translation units do not
    include it -- a drifted parameter list would silently pass mis-typed arguments.  Every `extern "C"` definition of
    csrc/*.hip is re-declared next to the header in one translation unit; C linkage forbids overloads, so g++ rejects
    any definition whose parameter types differ from the header's."""
    import tempfile
    decls, names = [], set()
    for f in sorted((ROOT / "brainmagick_amd" / "csrc").glob("*.hip")):
        for m in re.finditer(r'extern\s+"C"\s+((?:const\s+)?[\w\*\s]+?)\s*\b(bm_\w+)\s*\(([^)]*)\)\s*\{',
                             f.read_text(), flags=re.S):
            decls.append(f"{m.group(1).strip()} {m.group(2)}({' '.join(m.group(3).split())});")
            names.add(m.group(2))
    header = (ROOT / "include" / "bm_hip.h").read_text()
    declared = set(re.findall(r"^(?:int|long|const char\*|void)\s+(bm_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 85 and declared <= names, sorted(declared - names)
    with tempfile.TemporaryDirectory() as tmp:
        tu = Path(tmp) / "check.cpp"
        tu.write_text('#include "bm_hip.h"\nextern "C" {\n' + "\n".join(decls) + "\n}\nint main() { return 0; }\n")
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", f"-I{ROOT / 'include'}", str(tu)],
                           capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:2000]


# -- the reference Solver's contract, read from its source ----------------------------------------------------------
REFERENCE_BM = Path("/root/reference/bm")

# function -> {root expression (as source) -> role}: which local names / attributes of the reference's own code hold
# the objects that brainmagick_amd replaces
_CONTRACT_ROOTS = {
    ("solver.py", "Solver._process_batch"): {"self.model": "model", "self.feature_model": "feature_model",
                                             "self.scale_reject": "scale_reject", "batch": "batch"},
    ("solver.py", "Solver._run_one_epoch"): {"self.model": "model", "self.loss": "loss", "self.optimizer": "optimizer",
                                             "self.all_models": "module_list", "batch": "batch"},
    ("solver.py", "Solver._create_loss"): {"loss": "loss", "self.optimizer": "optimizer", "ClipLoss": "loss_class"},
    ("solver.py", "Solver.train"): {"self.scale_reject": "scale_reject"},
    ("solver.py", "Solver.predict"): {"SegmentBatch": "batch_class"},
    ("wer.py", "get_wer"): {"solver.model": "model", "solver.loss": "loss", "clip": "loss", "batch": "batch"},
}


def _reference_touches():
    """{role: {attribute or "__call__": [(n positional, (keyword names))]}} -- every attribute the reference's hot
    loop reads on, and every call it makes to, the objects in _CONTRACT_ROOTS, by walking the AST of the reference
    sources (nothing is imported: flashy / dora / julius are absent from this image)."""
    import ast
    touches = {}

    def note(role, name, call=None):
        calls = touches.setdefault(role, {}).setdefault(name, [])
        if call is not None and call not in calls:
            calls.append(call)

    trees = {}
    for (fname, qual), roots in _CONTRACT_ROOTS.items():
        tree = trees.setdefault(fname, ast.parse((REFERENCE_BM / fname).read_text()))
        node = tree
        for part in qual.split("."):
            node = next(n for n in ast.walk(node) if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name == part)
        for n in ast.walk(node):
            if isinstance(n, ast.Call):
                src = ast.unparse(n.func)
                sig = (len(n.args), tuple(sorted(k.arg for k in n.keywords if k.arg)),
                       any(k.arg is None for k in n.keywords))
                if src in roots:
                    note(roots[src], "__call__", sig)
                elif isinstance(n.func, ast.Attribute) and ast.unparse(n.func.value) in roots:
                    note(roots[ast.unparse(n.func.value)], n.func.attr, sig)
            elif isinstance(n, ast.Attribute) and ast.unparse(n.value) in roots:
                note(roots[ast.unparse(n.value)], n.attr)
            elif isinstance(n, ast.Call) is False and isinstance(n, ast.Name) and False:
                pass
        # len(batch...) / `if self.scale_reject:` style uses
        for n in ast.walk(node):
            if isinstance(n, ast.Call) and ast.unparse(n.func) == "len" and n.args and ast.unparse(n.args[0]) in roots:
                note(roots[ast.unparse(n.args[0])], "__len__")
    return touches


@pytest.mark.skipif(not REFERENCE_BM.exists(), reason="needs the reference checkout (build container only)")
def test_reference_solver_contract_is_met_by_the_replacement_classes():
    """INTEGRATION.md section 4 says the maintainer changes imports and one factory line and `bm/solver.py` /
    `bm/wer.py` stay as they are.  flashy / dora are absent, so the reference Solver cannot RUN here; this is the next
    best thing: every attribute `bm/solver.py:230-321, 343-394` and `bm/wer.py:21-121` touch on the model, the loss,
    the optimizer, the batch and ScaleReject is looked up on the replacement classes, and every call is bound against
    the replacement's signature with the reference's own argument count and keyword names."""
    import inspect
    from brainmagick_amd import synthetic
    from brainmagick_amd.models import SimpleConv
    from brainmagick_amd.losses import ClipLoss
    from brainmagick_amd.norm import ScaleReject
    from brainmagick_amd.optim import FlatAdam
    touches = _reference_touches()
    # the walk found the calls the hot loop is made of (a reference that moved would fail here, not pass vacuously)
    assert (2, (), False) in touches["model"]["__call__"]                     # self.model(inputs, batch)
    assert "modules" in touches["model"] and "eval" in touches["model"]
    assert (3, (), False) in touches["loss"]["__call__"]                      # self.loss(estimate, output, features_mask)
    assert {"train", "eval", "parameters", "get_probabilities", "to"} <= set(touches["loss"]) | {"to"}
    assert {"zero_grad", "step", "add_param_group"} <= set(touches["optimizer"])
    assert {"to", "meg", "features", "features_mask", "replace"} <= set(touches["batch"])
    assert "rejection_rate" in touches["scale_reject"] and "__call__" in touches["scale_reject"]
    assert "train" in touches["module_list"] and "state_dict" in touches["module_list"]

    batch = synthetic.make_batch(3, 6, 16, 4, 2, seed=0)
    model = SimpleConv(in_channels={"meg": 6}, out_channels=4, hidden={"meg": 8}, n_subjects=2, depth=2,
                       merger=False, subject_layers=False)
    loss = ClipLoss()
    # (FlatAdam and ScaleReject as classes: their instances need a GPU / a fitted scaler)
    from brainmagick_amd.models import DeepMel
    providers = {"model": model, "feature_model": DeepMel(4, 8, 2, 4, kernel=3, stride=1), "loss": loss, "optimizer": FlatAdam, "batch": batch,
                 "scale_reject": ScaleReject, "module_list": torch.nn.ModuleList([model]),
                 "loss_class": ClipLoss, "batch_class": type(batch)}

    def bind(fn, sig, what):
        npos, kws, has_star = sig
        try:
            inspect.signature(fn).bind(*([None] * npos), **{k: None for k in kws})
        except TypeError as exc:
            if has_star and not kws:
                return                      # `ClipLoss(**kw, ...)`: the keyword names are configuration, checked below
            raise AssertionError(f"{what}: the reference calls it with {npos} positional + {kws}: {exc}")

    # the wav2vec2 branch of bm/solver.py:304-316 (`feature_model.device`, `feature_model(output, output_dim=...)`)
    # belongs to bm/features' fine-tuned wav2vec2 model, outside SURVEY section 8; the DeepMel branch (:318) is the one
    # the replacement serves
    touches["feature_model"].pop("device", None)
    touches["feature_model"]["__call__"] = [c for c in touches["feature_model"]["__call__"] if "output_dim" not in c[1]]
    assert (1, (), False) in touches["feature_model"]["__call__"]
    for role, attrs in touches.items():
        obj = providers[role]
        for name, calls in attrs.items():
            if name == "__call__":
                target = obj if inspect.isclass(obj) else (obj.forward if isinstance(obj, torch.nn.Module) else obj.__call__)
                if role == "scale_reject":
                    target = ScaleReject.__call__
                    calls = [(c[0] + 1, c[1], c[2]) for c in calls]           # unbound: + self
                for sig in calls:
                    bind(target, sig, f"{role}(...)")
                continue
            assert hasattr(obj, name), f"the reference touches `{role}.{name}`; {type(obj).__name__} has no such attribute"
            for sig in calls:
                member = getattr(obj, name)
                if inspect.isclass(obj) and inspect.isfunction(member):
                    sig = (sig[0] + 1, sig[1], sig[2])                        # unbound: + self
                if callable(member) and not isinstance(member, torch.Tensor):
                    bind(member, sig, f"{role}.{name}(...)")
    # `ClipLoss(**args.clip, dset_args=args.dset)` (bm/solver.py:85-88): every key of conf/config.yaml's `clip:` block
    # minus the two the Solver pops must be a constructor argument
    import yaml
    clip_cfg = yaml.safe_load((REFERENCE_BM / "conf" / "config.yaml").read_text())["clip"]
    kw = {k: v for k, v in clip_cfg.items() if k not in ("save_best", "sync_grad")}
    inspect.signature(ClipLoss).bind(**kw, dset_args=None)
    # SegmentBatch(meg, features, mask, subjects, recordings) positionally (bm/solver.py `predict`)
    for sig in touches["batch_class"]["__call__"]:
        bind(type(batch), sig, "SegmentBatch(...)")


def test_bench_reads_the_counter_passes_like_the_guide_prescribes(tmp_path):
    """bench.py measures `roofline.traffic` in its own run from two rocprofv3 --pmc children: per kernel, bytes =
    (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (KiB units; FETCH_SIZE halves wide coalesced reads on gfx950)."""
    import bench
    rows = "Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n"
    for d, (k, v) in enumerate([("void conv_nn_h2w_kernel<3, 5>(ConvH2Args)", 100.0), ("void conv_nn_h2w_kernel<3, 5>(ConvH2Args)", 300.0),
                                ("affine_act_res_kernel<4>(float const*, float*)", 50.0)]):
        rows += f'{d},"{k}",FETCH_SIZE,{v}\n'
    (tmp_path / "a").mkdir()
    (tmp_path / "a" / "bench_counter_collection.csv").write_text(rows)
    fetch = bench.read_counter_csvs(tmp_path, "FETCH_SIZE")
    assert fetch == {"conv_nn_h2w_kernel<3, 5>": [2, 400.0], "affine_act_res_kernel<4>": [1, 50.0]}
    assert bench.read_counter_csvs(tmp_path, "WRITE_SIZE") is None
    write = {"conv_nn_h2w_kernel<3, 5>": [2, 100.0], "affine_act_res_kernel<4>": [1, 60.0]}
    per_launch, n, step_bytes, top = bench.traffic_from_counters(fetch, write, "conv_nn_h2w_kernel<3,5>", child_steps=1)
    assert n == 2 and per_launch == (2 * 200.0 + 50.0) * 1024
    assert step_bytes == 2 * per_launch + (2 * 50.0 + 60.0) * 1024
    assert list(top)[0] == "conv_nn_h2w_kernel<3, 5>"
    # the label covers every instantiation of the kernel (round 6: <3, 5, false> and the residual-epilogue <3, 5, true>):
    # launch-weighted bytes per launch
    fetch2 = {"conv_nn_h2w_kernel<3, 5, false>": [3, 300.0], "conv_nn_h2w_kernel<3, 5, true>": [1, 200.0]}
    write2 = {"conv_nn_h2w_kernel<3, 5, false>": [3, 150.0], "conv_nn_h2w_kernel<3, 5, true>": [1, 80.0]}
    per_launch, n, step_bytes, _ = bench.traffic_from_counters(fetch2, write2, "conv_nn_h2w_kernel<3,5>", child_steps=1)
    assert n == 4 and per_launch == pytest.approx(((2 * 300.0 + 150.0) + (2 * 200.0 + 80.0)) * 1024 / 4)
    assert step_bytes == pytest.approx(per_launch * 4)


def test_comm_timer_is_inert_without_a_gpu_and_names_the_phases():
    from brainmagick_amd import distrib
    t = distrib.CommTimer()
    distrib.set_comm_timer(t)
    try:
        with distrib._phase("reduce_scatter"):       # no GPU here: a null context, nothing recorded
            pass
        assert t.records == [] and t.summary(3) == {}
    finally:
        distrib.set_comm_timer(None)
    assert distrib.reported_world() == 1
