"""Helpers shared by the component GPU test files (tests/test_gpu_*.py): device guard, error measures, parity logging, the
network / oracle runners and the fixtures of the kernel cases.  Everything here was module-level code of the per-round files
(test_gpu_parity / round2 / round3 / round4) that round 5 re-filed by component."""
import collections
import ctypes
import json
import numpy as np
import os
import pytest
import socket
import subprocess
import sys
import torch
import torch.nn.functional as F
# ------------------------------------------------------------------------------------ whole network
import functools
from m3dssd_amd import _hip, synth
from m3dssd_amd import synth
from torch import nn


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("no ROCm device visible: the gpu-marked tests must run on the MI355X box")
    return torch.device("cuda:0")


def _relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


# ------------------------------------------------------------------------------------ conv
CONV_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad, bias, bn, act, res
    (2, 16, 20, 24, 16, 3, 1, 1, False, True, 1, False),      # level0-like, BK=16 path
    (2, 16, 20, 24, 32, 3, 2, 1, False, True, 1, False),      # level1-like, stride 2
    (1, 32, 18, 22, 64, 3, 2, 1, True, True, 1, False),       # tree conv1 stride 2
    (1, 64, 17, 19, 64, 3, 1, 1, True, True, 1, True),        # conv2 + residual, ragged M
    (1, 128, 16, 40, 128, 3, 1, 1, True, True, 1, True),
    (2, 256, 8, 20, 256, 3, 1, 1, True, True, 1, True),
    (1, 448, 16, 40, 128, 1, 1, 0, False, True, 1, False),    # root conv over a 448-ch concat
    (1, 128, 16, 40, 256, 1, 1, 0, True, True, 1, False),     # head layer 1
    (1, 128, 9, 11, 27, 3, 1, 1, True, False, 0, False),      # offset/mask conv (Cout 27)
    (3, 512, 4, 10, 512, 3, 1, 1, True, True, 1, True),       # level5-like, small M
]


def gpu_nms_empty():
    from lib.nms.gpu_nms import gpu_nms
    return gpu_nms(np.zeros((0, 5), dtype=np.float32), 0.4)


@functools.lru_cache(maxsize=None)
def _run_both(crop, B, pad):
    from model.M3d_inference_align import build
    from oracle import model_cpu
    dev = _dev()
    conf = synth.synth_conf(crop, 0, batch_size=B, device="cuda:0")
    sd = synth.synth_state_dict(0)
    x = synth.synth_frames(B, crop, 1234, pad_right_third=pad)
    net = build(conf, "test")
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    with torch.no_grad():
        out = net(x.to(dev))
    eng = net.engine()
    plan = eng.plan_for(B, crop[0], crop[1])
    fh, fw = crop[0] // 8, crop[1] // 8
    ind = plan.named["sel_idx"].view(B, 1, fh, fw).long().cpu()
    prob_sel = plan.named["sel_prob"].view(B, 1, fh, fw).cpu()
    # oracle free-running (its own decisions) and oracle with the engine's decisions injected
    cconf = synth.synth_conf(crop, 0, batch_size=B, device="cpu")
    taps_free, taps_inj = {}, {}
    with torch.no_grad():
        free = model_cpu.rpn_forward(sd, cconf, x, taps_free)
        inj = model_cpu.rpn_forward(sd, cconf, x, taps_inj,
                                    inject={"sel": {"ind": ind, "hard": (prob_sel > 0.5).float()}})
    return net, plan, out, free, inj, taps_free, taps_inj, ind, prob_sel


def _check_decisions(taps_free, ind, prob_sel):
    """Engine decisions == oracle decisions except at near-ties of the oracle's own fg probabilities."""
    fg = taps_free["fg_prob"]
    o_mask, o_ind = fg.max(dim=1, keepdim=True)
    diff = (o_ind != ind)
    if diff.any():
        alt = torch.gather(fg, 1, ind)
        assert ((o_mask - alt)[diff].abs() < 1e-4).all(), "top-1 anchor differs beyond a near-tie"
    assert (prob_sel - torch.gather(fg, 1, ind)).abs().max().item() < 1e-4
    hard_o, hard_e = (o_mask > 0.5), (prob_sel > 0.5)
    flip = hard_o != hard_e
    if flip.any():
        assert ((o_mask - 0.5)[flip].abs() < 1e-4).all(), "hard mask differs beyond a near-tie at 0.5"
    return int(diff.sum()), int(flip.sum())


def _clean_rows(taps_free, ind, prob_sel, A, radius=4):
    """Row mask [B, A*fh*fw]: rows whose pixel lies at least `radius` pixels away from every pixel where the engine and the
    free-running oracle took a different discrete decision (top-1 anchor / hard mask at an exact near-tie).  A differing
    decision changes that pixel's alignment offsets; center_align then resamples the aligned map around each pixel, so the
    neighbourhood is excluded too.  (z3d also passes through ANAB's global pooling: bounded separately by the callers.)"""
    fg = taps_free["fg_prob"]
    o_mask, o_ind = fg.max(dim=1, keepdim=True)
    bad = ((o_ind != ind) | ((o_mask > 0.5) != (prob_sel > 0.5))).float()
    if bad.any():
        bad = F.max_pool2d(bad, 2 * radius + 1, stride=1, padding=radius)
    ok = (bad == 0).view(bad.shape[0], 1, -1).expand(-1, A, -1).reshape(bad.shape[0], -1)
    return ok


def _parity_log(name, payload):
    import json
    d = os.path.join(os.path.dirname(GOLDEN.rstrip("/")), "..", "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_r06.jsonl"), "a") as f:
            f.write(json.dumps({"test": name, **payload}) + "\n")


# ------------------------------------------------------------------------------------ fused head + graph
def _head_case(seed, cin, cout, cpad, dev, n=2, h=13, w=21):
    """One 3-/2-layer head: returns (MlpDesc, device output, torch reference, keep-alive list)."""
    from m3dssd_amd import _hip
    from m3dssd_amd.engine import pack_frag
    from m3dssd_amd.host import standalone as S
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g)
    layers = []
    ref = x
    chans = ([cin, 256] if cin == 128 else [256]) + [256, cout]
    for li in range(len(chans) - 1):
        wt = torch.randn(chans[li + 1], chans[li], 1, 1, generator=g) / chans[li] ** 0.5
        b = torch.randn(chans[li + 1], generator=g) * 0.1
        last = li == len(chans) - 2
        bn = None
        ref = F.conv2d(ref, wt, b)
        if not last:
            bn = torch.nn.BatchNorm2d(chans[li + 1]).eval()
            with torch.no_grad():
                bn.weight.uniform_(0.5, 1.5, generator=g)
                bn.bias.normal_(0, 0.2, generator=g)
                bn.running_mean.normal_(0, 0.2, generator=g)
                bn.running_var.uniform_(0.5, 1.5, generator=g)
            ref = F.leaky_relu(bn(ref), 0.01)
        layers.append((wt, b, bn, last))
    v, _ = S._to_nhwc(x.to(dev))
    d = _hip.MlpDesc()
    keep = [v]
    d.inp, d.in_cs, d.M, d.Cin = v.ptr, v.cs, n * h * w, cin
    slots = ["1", "2", "3"] if cin == 128 else ["2", "3"]
    for slot, (wt, b, bn, last) in zip(slots, layers):
        co = wt.shape[0]
        wp = pack_frag(wt.reshape(co, wt.shape[1]), cpad if last else 256, dev)
        sc, sh = S._affine(co, b.to(dev), None if bn is None else bn.to(dev), dev)
        keep += [wp, sc, sh]
        setattr(d, "w" + slot, wp.data_ptr())
        setattr(d, "s" + slot, sc.data_ptr())
        setattr(d, "t" + slot, sh.data_ptr())
    out = torch.zeros(n, cout, h * w, device=dev)
    d.Cout, d.Cout_pad, d.out, d.out_img_stride, d.HW = cout, cpad, out.data_ptr(), cout * h * w, h * w
    return d, out, ref.detach(), keep


WINO_CASES = [
    # N, Cin, H, W, Cout, bias, bn, act, res, sigmoid_from
    (2, 16, 20, 24, 16, False, True, 1, False, -1),       # level0-like (one k-step)
    (1, 64, 18, 22, 64, True, True, 1, True, -1),         # ragged tile count (99 tiles), residual
    (1, 128, 16, 40, 128, True, True, 1, True, -1),
    (2, 256, 8, 20, 256, True, True, 1, False, -1),
    (1, 128, 10, 12, 27, True, False, 0, False, 18),      # offset/mask conv: Cout 27, sigmoid on the mask channels
    (3, 512, 4, 10, 512, True, True, 1, True, -1),
]


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _log(name, payload):
    """Measured margins of the parity tests, merged back from the GPU box (gpurun_out/parity_r06.jsonl)."""
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_r06.jsonl"), "a") as f:
            f.write(json.dumps({"test": name, **payload}) + "\n")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ------------------------------------------------------------------------------------ top-k select + decode
def _topk_inputs(R, A, seed, scores):
    g = torch.Generator().manual_seed(seed)
    prob = torch.rand(1, R, 4, generator=g) * 0.2
    prob[0, :, 1] = scores                     # class 1 carries the row score; the others stay below it
    prob[0, :, 2:] = prob[0, :, 2:] * 0.0 + (scores[:, None] * 0.5)
    b2 = torch.randn(1, R, 4, generator=g) * 0.3
    b3 = torch.randn(1, R, 7, generator=g) * 0.3
    x1 = torch.rand(R, generator=g) * 1000
    y1 = torch.rand(R, generator=g) * 300
    rois = torch.stack([x1, y1, x1 + 20 + torch.rand(R, generator=g) * 80, y1 + 20 + torch.rand(R, generator=g) * 60,
                        torch.randint(0, A, (R,), generator=g).float()], 1)
    anchors = torch.rand(A, 9, generator=g) * 10 + 1
    means, stds = torch.randn(11, generator=g) * 0.1, torch.rand(11, generator=g) + 0.5
    return prob, b2, b3, rois, anchors, means, stds


def _sortable_bits(score):
    u = score.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    bits = torch.where(u >= 0x80000000, (~u) & 0xFFFFFFFF, u | 0x80000000)
    return bits


TOPK_CASES = [
    ("random", 276480, 3000), ("ragged_R", 10007, 3000), ("k_equals_R", 1500, 1500), ("k1", 4097, 1), ("k_4096", 50000, 4096), ("k_10000", 60000, 10000), ("k_max", 50000, 16384),
    ("all_equal", 20000, 3000), ("16_levels", 276480, 3000), ("ties_at_cut", 30000, 3000), ("negative_and_zero", 9000, 2000),
]


# ------------------------------------------------------------------------------------ N > 1 on the leased GPU
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_RANK_SCRIPT = r"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, %(root)r)
from m3dssd_amd import dist as mdist, synth
from m3dssd_amd.host.detect import detect_device, select_block
from model.M3d_inference_align import build
rank, world, local = mdist.init_from_env(backend=%(backend)r)
local = local %% torch.cuda.device_count()
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
B = 4
conf = synth.synth_conf((128, 320), 0, batch_size=B // world, device=str(dev))
net = build(conf, "test")
net.load_state_dict(synth.synth_state_dict(0))
net = net.to(dev)
x = synth.synth_frames(B, (128, 320), 31)
mine = mdist.shard_batch(x, rank, world).to(dev)
block, counts = select_block(*detect_device(net, mine, conf), conf)
dets, cnt = mdist.gather_block(block)
torch.cuda.synchronize()
np.savez(os.path.join(%(out)r, "rank%%d.npz" %% rank), dets=dets.cpu().numpy(), counts=cnt.cpu().numpy())
torch.distributed.barrier()
torch.distributed.destroy_process_group()
"""


def _run_ranks(tmp_path, backend, world=2):
    import signal
    import subprocess
    import sys
    import types
    script = tmp_path / ("rank_%s.py" % backend)
    script.write_text(_RANK_SCRIPT % dict(root=ROOT, backend=backend, out=str(tmp_path)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    # own session: on a hang the whole process group (launcher + ranks) is killed, nothing is left holding the GPU
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        out, _ = p.communicate(timeout=240)
        return types.SimpleNamespace(returncode=p.returncode, stdout=out, timed_out=False)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, _ = p.communicate()
        return types.SimpleNamespace(returncode=-9, stdout=out or "", timed_out=True)


def _single_process_reference():
    from lib.rpn_util import detect_batch
    from model.M3d_inference_align import build
    dev = _dev()
    conf = synth.synth_conf((128, 320), 0, batch_size=2, device="cuda:0")
    net = build(conf, "test")
    net.load_state_dict(synth.synth_state_dict(0))
    net = net.to(dev)
    x = synth.synth_frames(4, (128, 320), 31).to(dev)
    # the same shard-sized batches the two ranks run (a batch-4 plan may split K differently: equal only to fp32 roundoff)
    parts = [tuple(t.clone() for t in detect_batch(net, x[lo:lo + 2], conf)) for lo in (0, 2)]
    d, c = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
    return d.cpu().numpy(), c.cpu().numpy()


CROP = (128, 320)


def _relerr_t(a, b):
    return ((a - b).abs() / (1 + b.abs())).max().item()


def _net_sd(seed=0, bs=2):
    from model.M3d_inference_align import build
    conf = synth.synth_conf(CROP, 0, batch_size=bs, device="cuda:0")
    net = build(conf, "test")
    return net, conf, synth.synth_state_dict(seed)


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


def _net_dev(seed=0, bs=2):
    from model.M3d_inference_align import build
    conf = synth.synth_conf(CROP, 0, batch_size=bs, device="cuda:0")
    net = build(conf, "test")
    net.load_state_dict(synth.synth_state_dict(seed))
    return net.to(_dev()), conf


def _w44_case(cin, cout, H, W, B, seed=0, res=False):
    from m3dssd_amd.engine import pack_wino44
    dev = _dev()
    g = torch.Generator().manual_seed(seed + cin + H)
    xf = torch.randn(B, cin, H, W, generator=g)
    wf = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    rf = torch.randn(B, cout, H, W, generator=g) if res else None
    x = xf.permute(0, 2, 3, 1).contiguous().to(dev)
    U = pack_wino44(wf, cout, dev)
    out = torch.zeros(B, H, W, cout, device=dev)
    r = rf.permute(0, 2, 3, 1).contiguous().to(dev) if res else None
    d = _hip.ConvDesc()
    d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.data_ptr(), cin, B, H, W, cin
    d.wgt, d.Cout, d.Cout_pad = U.data_ptr(), cout, cout
    d.kh = d.kw = 3
    d.stride, d.pad, d.dil, d.Ho, d.Wo = 1, 1, 1, H, W
    d.out, d.out_cs, d.act, d.sigmoid_from = out.data_ptr(), cout, 1, -1
    if res:
        d.res, d.res_cs, d.res_mode = r.data_ptr(), cout, 0
    ref = F.conv2d(xf.double(), wf.double(), padding=1)
    if res:
        ref = ref + rf.double()
    ref = torch.where(ref > 0, ref, ref * 0.01).float().permute(0, 2, 3, 1)
    return d, out, ref, (x, U, r)


def _net_dt(crop, B, dtype="f32"):
    from model.M3d_inference_align import build
    conf = synth.synth_conf(crop, 0, batch_size=B, device="cuda:0")
    net = build(conf, "test")
    net.load_state_dict(synth.synth_state_dict(0))
    return net.to(_dev()).set_compute_dtype(dtype), conf


__all__ = ['CONV_CASES', 'CROP', 'GOLDEN', 'ROOT', 'TOPK_CASES', 'WINO44_CASES', 'WINO_CASES', '_RANK_SCRIPT', '_check_decisions', '_clean_rows', '_dev', '_free_port', '_head_case', '_log', '_net_dev', '_net_dt', '_net_sd', '_parity_log', '_relerr', '_relerr_t', '_run_bench', '_run_both', '_run_ranks', '_single_process_reference', '_soak', '_sortable_bits', '_stream', '_topk_inputs', '_w44_case', 'gpu_nms_empty']
