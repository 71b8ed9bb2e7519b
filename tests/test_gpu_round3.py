"""GPU tests added in round 3 (all through the C ABI of libm3dssd_hip.so):

  * the host boundary on a device: 'module.'-prefixed checkpoints, loads through wrappers / containers re-pack the engine,
    nn.DataParallel(net)(x) on the leased device == net(x) (scripts/test_rpn_3d.py:50-54);
  * bench.py's N > 1 branch executed end to end with two ranks (M3D_DIST_BACKEND=gloo: the lease has ONE device, RCCL refuses
    duplicate devices) -- the code path the driver runs on the 8-GPU node;
  * the DCN / DCNv2 module contract with deformable_groups > 1 (model/DCNv2/test.py:169-179) against the oracle;
  * non-finite sampling positions behave like the reference's compares (a NaN coordinate fails `h_im > -1 && ...`).
"""
import collections
import json
import os
import subprocess
import sys

import pytest
import torch
from torch import nn

from m3dssd_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CROP = (128, 320)


def _dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def _relerr(a, b):
    return ((a - b).abs() / (1 + b.abs())).max().item()


def _net(seed=0, bs=2):
    from model.M3d_inference_align import build
    conf = synth.synth_conf(CROP, 0, batch_size=bs, device="cuda:0")
    net = build(conf, "test")
    return net, conf, synth.synth_state_dict(seed)


# ------------------------------------------------------------------------------------ host boundary
def test_module_prefixed_checkpoint_then_forward_equals_plain_load():
    dev = _dev()
    x = synth.synth_frames(2, CROP, 5).to(dev)
    net, conf, sd = _net()
    net.load_state_dict(sd)
    ref = [t.clone() for t in net.to(dev)(x)[:4]]
    net2, _, _ = _net()
    net2.load_state_dict(collections.OrderedDict(("module." + k, v) for k, v in sd.items()), strict=True)
    got = net2.to(dev)(x)[:4]
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


def test_load_through_wrapper_repacks_the_engine():
    """forward -> wrapper.load_state_dict(other checkpoint) -> forward must run the NEW weights (nn.Module.load_state_dict on a
    parent recurses through child._load_from_state_dict and never calls RPN.load_state_dict)."""
    dev = _dev()
    x = synth.synth_frames(2, CROP, 6).to(dev)
    net, conf, sd0 = _net(0)
    sd1 = synth.synth_state_dict(1)
    net.load_state_dict(sd0)
    net = net.to(dev)
    out0 = [t.clone() for t in net(x)[:4]]
    wrapper = nn.DataParallel(net, device_ids=[0])
    wrapper.load_state_dict(collections.OrderedDict(("module." + k, v) for k, v in sd1.items()))
    out1 = [t.clone() for t in net(x)[:4]]
    assert not torch.equal(out0[3], out1[3]), "the packed engine still holds the previous checkpoint"
    fresh, _, _ = _net(1)
    fresh.load_state_dict(sd1)
    ref1 = fresh.to(dev)(x)[:4]
    for a, b in zip(out1, ref1):
        assert torch.equal(a, b)
    holder = nn.ModuleDict({"det": net})
    holder.load_state_dict(collections.OrderedDict(("det." + k, v) for k, v in sd0.items()))
    for a, b in zip(net(x)[:4], out0):
        assert torch.equal(a, b)


def test_data_parallel_wrapper_on_the_leased_device_equals_the_module():
    """The reference script wraps the net in nn.DataParallel (scripts/test_rpn_3d.py:50-51).  With one visible device the
    wrapper calls the module itself; a replica made for a second device would pack its own engine (the replica hook is
    exercised directly: it must not share the source's plans)."""
    dev = _dev()
    x = synth.synth_frames(2, CROP, 7).to(dev)
    net, conf, sd = _net()
    net.load_state_dict(sd)
    net = net.to(dev)
    ref = [t.clone() for t in net(x)]
    got = nn.DataParallel(net, device_ids=[0])(x)
    assert len(got) == 6
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    # a REAL replica (torch.nn.parallel.replicate: `_parameters` is empty, the broadcast copies hang on it as plain attributes
    # and every child is a replica too) must run: it asks the source module for the engine of its device
    from torch.nn.parallel import replicate
    rep = replicate(net, [0])[0]
    assert rep is not net and len(list(rep.parameters())) == 0 and getattr(rep, "_is_replica", False)
    assert rep._engine is None and net._engine is not None
    for a, b in zip(rep(x)[:4], ref[:4]):
        assert torch.equal(a, b)
    assert rep.engine(x.device) is net._engine          # same device: the source's own packed engine, not a second copy
    # a replica "on another device" packs a per-device engine once, from the source's state_dict, and keeps it across replicas
    e1 = net._engine_for("cuda:0")
    assert e1 is net._engine
    net.refresh_engine()
    assert "_device_engines" not in net.__dict__ and net._engine is None


# ------------------------------------------------------------------------------------ bench.py, N > 1 branch
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_bench(extra, env_extra, timeout=600):
    import signal
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", **env_extra)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, start_new_session=True, cwd=ROOT)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        raise AssertionError("bench.py timed out\n" + (err or "")[-3000:])
    return p.returncode, out, err


def test_bench_two_ranks_gloo_emits_one_parseable_line():
    """`python bench.py --gpus 2` self-launches under torch.distributed.run; both ranks share the leased device (gloo), run the
    pipelined graph, the per-step gather_block, the barriers and the all_reduce(MAX) of the elapsed time; rank 0 prints ONE
    JSON line for the whole job."""
    rc, out, err = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                              {"M3D_DIST_BACKEND": "gloo"})
    assert rc == 0, err[-3000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 16 and r["config"]["per_gpu_batch"] == 8
    assert r["steps"] == 3 and r["scaling"] == "weak" and r["dist_backend"] == "gloo"
    assert r["value"] > 0 and abs(r["value"] - 16 * 3 / (r["ms_per_step"] * 3e-3)) / r["value"] < 0.01
    assert "configs2_bf16" not in r and "cpu_baseline" not in r          # N = 1 only
    assert r["roofline"]["frac"] > 0
    # the N > 1 line proves what the collective saw and isolates its cost (VERDICT r3 #4d)
    c = r["rccl"]
    assert c["world_size"] == 2 and c["backend"] == "gloo" and len(c["ranks_seen"]) == 2
    assert sorted(d["rank"] for d in c["ranks_seen"]) == [0, 1] and len({d["pid"] for d in c["ranks_seen"]}) == 2
    assert c["distinct_devices"] == 1                   # both test ranks share the leased GPU (RCCL would need 2: see below)
    assert c["gathered_rows"][0] == 16 and c["gathered_rows"][2] == 14 and c["allgather_us"] > 0
    assert c["shards_recomputed_on_rank0"] == 2 and c["shards_match"] is True
    assert r["step_roofline"]["mfma_frac"] > 0 and 0 < r["mfma_time_weighted_frac"] < 1
    assert r["helper_kernels"]["bundle"]["algorithmic_gbs"] > 0


def test_bench_nccl_refuses_more_ranks_than_devices():
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a single-device lease")
    rc, out, err = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], {"M3D_DIST_BACKEND": "nccl"},
                              timeout=300)
    assert rc != 0 and "visible devices" in (out + err)


def test_bench_default_line_carries_configs2_bf16():
    """The driver's command (`python bench.py`, N = 1): the f32 headline line also holds the bs = 64 bf16 measurement."""
    rc, out, err = _run_bench(["--steps", "5", "--warmup", "2", "--configs2-steps", "3", "--no-cpu-baseline"], {})
    assert rc == 0, err[-3000:]
    r = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert r["dtype"] == "f32" and r["config"]["per_gpu_batch"] == 8
    c2 = r["configs2_bf16"]
    assert c2["dtype"] == "bf16" and c2["config"]["per_gpu_batch"] == 64 and c2["steps"] == 3
    assert c2["value"] > r["value"] and c2["roofline"]["peak"] == 2500.0 and "step_roofline" in c2


# ------------------------------------------------------------------------------------ DCN module contract
@pytest.mark.parametrize("shape", [(2, 64, 32, 40, 64, 3, 1, 1, 2), (1, 48, 9, 11, 20, 3, 2, 1, 4), (2, 64, 16, 16, 32, 1, 1, 0, 2)])
def test_dcn_v2_deformable_groups_match_oracle(shape):
    """deformable_groups > 1 through the drop-in op: group g's channels sample at group g's offsets / masks
    (dcn_v2_im2col_cuda.cu:139-156)."""
    from m3dssd_amd.host import ops
    from oracle import dcn as odcn
    dev = _dev()
    n, c, h, w, co, k, stride, pad, G = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(co, c, k, k, generator=g) / (c * k * k) ** 0.5
    b = torch.randn(co, generator=g)
    ho, wo = odcn.out_size(h, w, k, k, stride, pad, 1)
    off = torch.randn(n, G * 2 * k * k, ho, wo, generator=g) * 2.0
    m = torch.rand(n, G * k * k, ho, wo, generator=g)
    ref = odcn.dcn_v2_forward(x, off, m, wt, b, stride, pad, 1, G)
    got = ops.dcn_v2_forward(x.to(dev), off.to(dev), m.to(dev), wt.to(dev), b.to(dev), stride, pad, 1, G).cpu()
    assert got.shape == ref.shape and _relerr(got, ref) < 2e-4
    # the groups really differ: feeding group 0's offsets to every group changes the result
    off0 = off[:, :2 * k * k].repeat(1, G, 1, 1)
    assert _relerr(odcn.dcn_v2_forward(x, off0, m, wt, b, stride, pad, 1, G), ref) > 1e-2
    with pytest.raises(RuntimeError):
        ops.dcn_v2_forward(x.to(dev), off.to(dev), m.to(dev), wt.to(dev), b.to(dev), stride, pad, 1, 5)


def test_dcn_module_example_of_the_reference_with_two_groups():
    """model/DCNv2/test.py:169-179 `example_dconv`: DCN(64, 64, (3, 3), 1, 1, deformable_groups=2) on 2 x 64 x 128 x 128."""
    from model.DCNv2.dcn_v2 import DCN
    from oracle import dcn as odcn
    dev = _dev()
    torch.manual_seed(3)
    dcn = DCN(64, 64, kernel_size=(3, 3), stride=1, padding=1, deformable_groups=2)
    assert tuple(dcn.conv_offset_mask.weight.shape) == (54, 64, 3, 3)
    with torch.no_grad():
        dcn.conv_offset_mask.weight.normal_(0, 0.02)
        dcn.conv_offset_mask.bias.normal_(0, 0.5)
        dcn.bias.normal_(0, 0.1)
    x = torch.randn(2, 64, 128, 128)
    with torch.no_grad():
        out = torch.nn.functional.conv2d(x, dcn.conv_offset_mask.weight, dcn.conv_offset_mask.bias, padding=1)
        o1, o2, mask = torch.chunk(out, 3, dim=1)                              # dcn_v2.py:65-68
        ref = odcn.dcn_v2_forward(x, torch.cat((o1, o2), 1), torch.sigmoid(mask), dcn.weight, dcn.bias, 1, 1, 1, 2)
        got = dcn.to(dev)(x.to(dev)).cpu()
    assert tuple(got.shape) == (2, 64, 128, 128)
    assert _relerr(got, ref) < 5e-4


def test_dcn_non_finite_sampling_positions_contribute_nothing():
    """`h_im > -1 && w_im > -1 && h_im < H && w_im < W` (dcn_v2_im2col_cuda.cu:165) is false for a NaN coordinate: the tap is
    skipped.  csrc/common.h dcn_corners decides with fminf / fmaxf (which drop NaNs) and therefore carries an explicit
    non-finite term; +-inf positions are outside by the ordinary test."""
    from m3dssd_amd.host import ops
    from oracle import dcn as odcn
    dev = _dev()
    g = torch.Generator().manual_seed(77)
    n, c, h, w, co, k = 1, 32, 8, 16, 32, 3
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(co, c, k, k, generator=g) / 17.0
    b = torch.randn(co, generator=g)
    off = torch.randn(n, 2 * k * k, h, w, generator=g)
    m = torch.rand(n, k * k, h, w, generator=g)
    nan, inf = float("nan"), float("inf")
    off[0, 0, 2, 3] = nan          # tap 0: dh NaN, dw finite
    off[0, 3, 2, 4] = nan          # tap 1: dw NaN, dh finite
    off[0, 8, 5, 5] = nan
    off[0, 9, 5, 5] = nan          # tap 4: both
    off[0, 10, 6, 6] = inf
    off[0, 13, 6, 7] = -inf
    ref = odcn.dcn_v2_forward(x, off, m, wt, b, 1, 1, 1, 1)
    assert torch.isfinite(ref).all()
    got = ops.dcn_v2_forward(x.to(dev), off.to(dev), m.to(dev), wt.to(dev), b.to(dev), 1, 1, 1, 1).cpu()
    assert torch.isfinite(got).all()
    assert _relerr(got, ref) < 2e-4


# ------------------------------------------------------------------------------------ run-to-run identity, every kernel family
def _soak(dtype, B, crop, n):
    """n forwards on the same frames: every output and every named intermediate buffer of the plan (all kernel families of the
    step write one: Winograd wave / LDS, wave-granular conv plain and deformable, block igemm, fused heads, halo tile, DCNv2
    patch / implicit GEMM, front end, ANAB, pooling / up-sampling helpers) compared bit for bit with the first forward."""
    from model.M3d_inference_align import build
    conf = synth.synth_conf(crop, 0, batch_size=B, device="cuda:0")
    net = build(conf, "test")
    net.load_state_dict(synth.synth_state_dict(0), strict=True)
    net = net.to(_dev()).set_compute_dtype(dtype)
    x = synth.synth_frames(B, crop, 7).to(_dev())
    ref, plan, bad, kinds = None, None, [], set()
    for it in range(n):
        with torch.no_grad():
            outs = net(x)[:4]
        torch.cuda.synchronize()
        if plan is None:
            plan = net.engine().plan_for(B, *crop)
            kinds = set(op[1].split("<")[0] for op in plan.ops)
        snap = {"out%d" % i: t.clone() for i, t in enumerate(outs)}
        for k, v in plan.named.items():
            t = getattr(v, "t", v)
            if torch.is_tensor(t):
                snap[k] = t.clone()
        if ref is None:
            ref = snap
            continue
        diff = [k for k in snap if not torch.equal(ref[k].view(torch.uint8), snap[k].view(torch.uint8))]
        if diff:
            bad.append((it, diff[:6]))
    return bad, len(ref), kinds


def test_fp32_forward_is_bit_identical_over_100_runs_at_the_bench_size():
    """the compare-built padding predicates of the non-deformable MFMA kernels (Winograd wave, plain wave conv,
    fused heads) sit next to matrix instructions at 2-3 waves per SIMD like the deformable kernels did when they dropped a
    corner once per 10^5..10^6 states; a wrong select there would show as run-to-run differences at these grid sizes."""
    bad, nbuf, kinds = _soak("f32", 8, (384, 1280), 100)
    assert {"wino44", "conv_wave", "head_mlp", "igemm"} <= kinds, kinds
    assert not bad, bad[:3]
    assert nbuf > 20


def test_fp32_forward_without_f4x4_is_bit_identical_over_40_runs(monkeypatch):
    """The F(2x2,3x3) wave kernel (and its split-K form) serves the 3x3 layers wherever the F(4x4) kernel does not apply or is
    switched off (M3D_WINO44=0): same soak on that plan."""
    import m3dssd_amd.engine as E
    monkeypatch.setattr(E, "USE_WINO44", False)
    bad, nbuf, kinds = _soak("f32", 8, (384, 1280), 40)
    assert "wino_wave" in kinds and "wino44" not in kinds, kinds
    assert not bad, bad[:3]


def test_bf16_forward_is_bit_identical_over_40_runs_at_batch_64():
    bad, nbuf, kinds = _soak("bf16", 64, (384, 1280), 40)
    assert {"bf16_halo", "bf16_conv", "bf16_head2", "bf16_frontend2", "bf16_anab", "bf16_dcn_patch"} <= kinds, kinds
    assert not bad, bad[:3]


# ------------------------------------------------------------------------------------ Winograd F(4x4,3x3)
WINO44_CASES = [
    # n, cin, h, w, cout, bias, bn, act, res
    (1, 16, 4, 4, 128, False, False, 0, False),          # one tile, one stage: every patch border is an image border
    (2, 32, 8, 12, 128, True, False, 1, False),          # 12 tiles: ragged 16-tile strip, strips crossing image rows
    (1, 128, 16, 40, 128, True, True, 1, True),          # level3 geometry (scaled): BN + residual + LeakyReLU
    (2, 64, 12, 20, 256, True, True, 1, False),          # two channel blocks of 128
    (1, 48, 20, 36, 100, True, True, 0, True),           # Cout 100 (pad 128), three stages
    (3, 128, 48, 160, 128, False, True, 1, True),        # full-size level3 map, 3 images
]


@pytest.mark.parametrize("nb", [1, 2])
@pytest.mark.parametrize("case", WINO44_CASES + [(2, 64, 24, 32, 64, True, True, 1, True), (1, 32, 8, 8, 40, True, False, 0, False)])
def test_winograd_f4x4_conv3x3_matches_torch(case, nb):
    """m3d_wino44_conv3x3_forward (Winograd F(4x4,3x3) on v_mfma_f32_16x16x4_f32, csrc/wino44_conv.hip) vs F.conv2d, same
    epilogue contract as the F(2x2,3x3) kernels.  Tolerance 2e-4 (1 + |ref|) like every fp32 conv here; the measured error is
    logged next to that of the F(2x2,3x3) wave kernel on the same operands (F(4x4) amplifies fp32 rounding ~7x)."""
    import torch.nn.functional as F
    from m3dssd_amd.host import standalone as S
    dev = _dev()
    n, ci, h, w, co, bias, bn, act, res = case
    if nb == 2 and co <= 64:
        nb = 1                                              # 64-channel layers: one 16-channel block per wave only
    g = torch.Generator().manual_seed(sum(case) + 11)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    b = torch.randn(co, generator=g) if bias else None
    ref = F.conv2d(x.double(), wt.double(), None if b is None else b.double(), padding=1)
    bnm = None
    if bn:
        bnm = torch.nn.BatchNorm2d(co).eval()
        with torch.no_grad():
            bnm.weight.uniform_(0.5, 1.5, generator=g)
            bnm.bias.normal_(0, 0.2, generator=g)
            bnm.running_mean.normal_(0, 0.2, generator=g)
            bnm.running_var.uniform_(0.5, 1.5, generator=g)
        ref = bnm.double()(ref)
        bnm = bnm.float()
    r = None
    if res:
        r = torch.randn(n, co, h, w, generator=g)
        ref = ref + r.double()
    if act:
        ref = F.leaky_relu(ref, 0.01)
    ref = ref.float()
    errs = {}
    for kind in ("wino44", "wino22"):
        if kind == "wino22" and (h % 2 or w % 2):
            continue
        with torch.no_grad():
            v, _ = S._to_nhwc(x.to(dev))
            rv = S._to_nhwc(r.to(dev))[0] if res else None
            out, keep = S.conv_nhwc(v, wt.to(dev), None if b is None else b.to(dev), None if bnm is None else bnm.to(dev), 1, 1,
                                    act=act, res=rv, wino44=(kind == "wino44"), wino44_nb=nb, wino=(kind == "wino22"), wino_variant=1)
            got = S._to_nchw(out, co).cpu()
        assert got.shape == ref.shape
        errs[kind] = ((got - ref).abs() / (1 + ref.abs())).max().item()
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_r04.jsonl"), "a") as f:
            f.write(json.dumps({"test": "wino44", "case": list(case), "nb": nb, **errs}) + "\n")
    assert errs["wino44"] < 2e-4, errs


@pytest.mark.parametrize("case,want", [((8, 256, 24, 80, 256, True, True, 1, True), 2),      # level4 at bs 8: 120 workgroups -> 240
                                       ((8, 512, 12, 40, 512, False, True, 1, True), 4),     # level5 at bs 8: 60 -> 240
                                       ((4, 128, 24, 80, 500, True, True, 0, False), 2),     # Cout 500 (pad 512), 64-channel slices
                                       ((1, 256, 8, 8, 128, True, False, 1, True), 1)])      # too small to fill the chip: no split
def test_winograd_f4x4_splitk_matches_torch(case, want):
    """Split-K form of the F(4x4,3x3) kernel (K slices as gridDim.z + m3d_launch_splitk_reduce in slice order): the plan, the
    result vs F.conv2d in fp64, and bitwise repeatability (no atomics)."""
    import ctypes
    import torch.nn.functional as F
    from m3dssd_amd import _hip
    from m3dssd_amd.host import standalone as S
    dev = _dev()
    n, ci, h, w, co, bias, bn, act, res = case
    g = torch.Generator().manual_seed(sum(case) + 5)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    b = torch.randn(co, generator=g) if bias else None
    ref = F.conv2d(x.double(), wt.double(), None if b is None else b.double(), padding=1)
    bnm = None
    if bn:
        bnm = torch.nn.BatchNorm2d(co).eval()
        with torch.no_grad():
            bnm.weight.uniform_(0.5, 1.5, generator=g)
            bnm.bias.normal_(0, 0.2, generator=g)
            bnm.running_mean.normal_(0, 0.2, generator=g)
            bnm.running_var.uniform_(0.5, 1.5, generator=g)
        ref = bnm.double()(ref)
        bnm = bnm.float()
    r = None
    if res:
        r = torch.randn(n, co, h, w, generator=g)
        ref = ref + r.double()
    if act:
        ref = F.leaky_relu(ref, 0.01)
    ref = ref.float()
    d = _hip.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Cout, d.Cout_pad = n, h, w, ci, co, -(-co // 128) * 128
    d.kh = d.kw = 3
    d.stride = d.pad = d.dil = 1
    d.Ho, d.Wo, d.in_cs, d.sigmoid_from = h, w, ci, -1
    splits, ws_bytes = ctypes.c_int(), ctypes.c_longlong()
    _hip.check(_hip.lib().m3d_wino44_splitk_plan(ctypes.byref(d), ctypes.byref(splits), ctypes.byref(ws_bytes)))
    assert splits.value == want and ws_bytes.value == (want * n * h * w * d.Cout_pad * 4 if want > 1 else 0)
    outs = []
    with torch.no_grad():
        v, _ = S._to_nhwc(x.to(dev))
        rv = S._to_nhwc(r.to(dev))[0] if res else None
        for _ in range(3):
            out, keep = S.conv_nhwc(v, wt.to(dev), None if b is None else b.to(dev), None if bnm is None else bnm.to(dev), 1, 1,
                                    act=act, res=rv, wino44=True, wino44_nb=2, wino_splitk=True)
            outs.append(S._to_nchw(out, co).cpu())
    err = ((outs[0] - ref).abs() / (1 + ref.abs())).max().item()
    assert err < 2e-4, err
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("shape", [(1, 4, 4), (2, 8, 12), (3, 20, 36), (1, 48, 160), (2, 384, 1280)])
def test_level0_winograd_f4x4_matches_torch(shape):
    """m3d_conv3x3_c16_wino (DLA level0, 3x3 16 -> 16 + folded BN + LeakyReLU as F(4x4,3x3) in three LDS phases) vs F.conv2d in
    fp64 and vs the direct kernel m3d_conv3x3_c16 on the same operands; one tile, ragged 16-tile strips, strips that cross image
    rows / images, the full 1280x384 frame."""
    import torch.nn.functional as F
    from m3dssd_amd import _hip
    from m3dssd_amd.engine import pack_wino44_c16
    L = _hip.lib()
    dev = _dev()
    n, h, w = shape
    g = torch.Generator().manual_seed(h * 7 + w)
    x = torch.randn(n, 16, h, w, generator=g)
    wt = torch.randn(16, 16, 3, 3, generator=g) / 12.0
    sc = torch.rand(16, generator=g) + 0.5
    sh = torch.randn(16, generator=g) * 0.2
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1), 0.01).float()
    xin = x.permute(0, 2, 3, 1).contiguous().to(dev)
    U = pack_wino44_c16(wt, dev)
    wd = wt.permute(2, 3, 1, 0).contiguous().to(dev)
    scd, shd = sc.to(dev), sh.to(dev)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for fn, wgt in ((L.m3d_conv3x3_c16_wino, U), (L.m3d_conv3x3_c16, wd)):
        out = torch.full((n, h, w, 16), 777.0, device=dev)
        _hip.check(fn(xin.data_ptr(), 16, wgt.data_ptr(), scd.data_ptr(), shd.data_ptr(), out.data_ptr(), 16, n, h, w, st))
        torch.cuda.synchronize()
        outs.append(out.permute(0, 3, 1, 2).cpu())
    e44 = ((outs[0] - ref).abs() / (1 + ref.abs())).max().item()
    edir = ((outs[1] - ref).abs() / (1 + ref.abs())).max().item()
    assert e44 < 2e-4 and edir < 2e-5, (e44, edir)
    again = torch.full((n, h, w, 16), 777.0, device=dev)
    _hip.check(L.m3d_conv3x3_c16_wino(xin.data_ptr(), 16, U.data_ptr(), scd.data_ptr(), shd.data_ptr(), again.data_ptr(), 16, n, h, w, st))
    torch.cuda.synchronize()
    assert torch.equal(again.permute(0, 3, 1, 2).cpu(), outs[0])


# ------------------------------------------------------------------------------------ NMS beyond the device reduce's 4096 rows
@pytest.mark.parametrize("n", [4097, 6000, 12000])
def test_nms_twin_has_no_row_limit(n):
    """`_nms` (lib/nms/gpu_nms.hpp:1-2) has no row limit in the reference (nms_kernel.cu:91-144).  The on-device greedy reduce holds
    4096 rows; larger inputs go through device masks + the reference's host pass and must give the oracle's keep list bit for bit."""
    import numpy as np
    from lib.nms.gpu_nms import gpu_nms
    from oracle import nms as onms
    rng = np.random.default_rng(n)
    ctr = rng.uniform([0, 0], [1280, 384], (n, 2))
    wh = rng.uniform([5, 5], [200, 150], (n, 2))
    score = rng.permutation(n).astype(np.float64) / n
    dets = np.concatenate([ctr - wh / 2, ctr + wh / 2, score[:, None]], 1).astype(np.float32)
    got = gpu_nms(dets, 0.4, device_id=0)
    want = onms.gpu_nms(dets, 0.4)
    assert list(got) == list(want) and len(got) > 50
