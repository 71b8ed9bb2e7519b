"""GPU tests added in round 4 (all through the C ABI of libm3dssd_hip.so):

  * the fed-input form of the pipelined detector: uint8 frames uploaded from pinned host memory on a copy stream, double
    buffered against the graph of the previous batch -- batch k's detections must equal detect_batch of frame set k
    (lib/rpn_util.py:1427-1429, lib/dataloader.py:934-950, lib/augmentations.py:472-501);
  * the 64-channel F(4x4) Winograd form at two workgroups per CU against a float64 convolution and against the
    one-workgroup build, bit for bit (same arithmetic, different occupancy).
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from m3dssd_amd import _hip, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CROP = (128, 320)


def _dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def _net(seed=0, bs=2):
    from model.M3d_inference_align import build
    conf = synth.synth_conf(CROP, 0, batch_size=bs, device="cuda:0")
    net = build(conf, "test")
    net.load_state_dict(synth.synth_state_dict(seed))
    return net.to(_dev()), conf


def test_fed_uint8_pipeline_equals_detect_batch_of_each_frame_set():
    from lib.rpn_util import detect_batch
    from m3dssd_amd.pipeline import PipelinedDetector
    dev = _dev()
    net, conf = _net()
    B, fh, fw = 2, 120, 310                                 # frames smaller than the crop: the stem pads them (Preprocess)
    rng = np.random.RandomState(3)
    sets = [torch.from_numpy(rng.randint(0, 256, size=(B, fh, fw, 3)).astype(np.uint8)).pin_memory() for _ in range(5)]
    want = []
    for fr in sets:
        d, c = detect_batch(net, fr.to(dev), conf)
        want.append((d.clone(), c.clone()))
    pipe = PipelinedDetector(net, conf, B, CROP[0], CROP[1], u8_frame=(fh, fw))
    got = []
    pipe.feed(sets[0])
    for k in range(len(sets)):
        if k + 1 < len(sets):
            pipe.feed(sets[k + 1])                          # upload of batch k + 1 overlaps the graph of batch k
        r = pipe.step_fed()
        if r is not None:
            got.append((r[0].clone(), r[1].clone()))
    r = pipe.flush()
    got.append((r[0].clone(), r[1].clone()))
    assert len(got) == len(sets)
    for k, ((d, c), (wd, wc)) in enumerate(zip(got, want)):
        assert torch.equal(c, wc), "batch %d: counts differ" % k
        assert torch.equal(d, wd), "batch %d: detections differ" % k
    assert any(int(c.sum()) > 0 for _, c in want)
    # the float input form of the same detector still works next to it and agrees with the uint8 path
    with pytest.raises(RuntimeError):
        pipe.feed(sets[0].to(torch.float32))
    pipe.feed(sets[0])
    pipe.feed(sets[1])
    with pytest.raises(RuntimeError):
        pipe.feed(sets[2])                                  # both buffers hold unsubmitted batches


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


@pytest.mark.parametrize("case", [(64, 64, 24, 80, 2, False), (128, 128, 16, 48, 2, True), (256, 256, 8, 20, 1, False),
                                  (32, 64, 12, 20, 3, True)])
def test_winograd_f4x4_two_workgroups_per_cu_matches_float64(case):
    cin, cout, H, W, B, res = case
    L = _hip.lib()
    d, out, ref, keep = _w44_case(cin, cout, H, W, B, res=res)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _hip.check(L.m3d_wino44_conv3x3_forward_ex(ctypes.byref(d), 1, st))
    torch.cuda.synchronize()
    err = ((out.cpu() - ref).abs() / (1 + ref.abs())).max().item()
    assert err < 2e-4, err
    first = out.clone()
    for _ in range(10):                                       # run to run: bit-identical
        out.zero_()
        _hip.check(L.m3d_wino44_conv3x3_forward_ex(ctypes.byref(d), 1, st))
        torch.cuda.synchronize()
        assert torch.equal(out, first)


@pytest.mark.parametrize("case", [(128, 128, 16, 48, 2, True), (256, 256, 24, 80, 2, False), (512, 64, 8, 12, 1, True)])
def test_winograd_f4x4_kpair_workgroups_match_float64_and_the_plain_form(case):
    """Layers too small to fill the CU slots run K-pair workgroups (512 threads, the two halves of the input channels side by
    side, accumulators traded through LDS): against float64, against the plain 64-channel form (summation order differs:
    tolerance), and run to run bit-identical."""
    cin, cout, H, W, B, res = case
    L = _hip.lib()
    d, out, ref, keep = _w44_case(cin, cout, H, W, B, res=res)
    assert L.m3d_wino44_kpair(ctypes.byref(d)) == 1
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _hip.check(L.m3d_wino44_conv3x3_forward_ex(ctypes.byref(d), 1, st))
    torch.cuda.synchronize()
    got = out.clone()
    err = ((got.cpu() - ref).abs() / (1 + ref.abs())).max().item()
    assert err < 2e-4, err
    for _ in range(10):
        out.zero_()
        _hip.check(L.m3d_wino44_conv3x3_forward_ex(ctypes.byref(d), 1, st))
        torch.cuda.synchronize()
        assert torch.equal(out, got)
    if cout % 128 == 0:
        _hip.check(L.m3d_wino44_conv3x3_forward_ex(ctypes.byref(d), 2, st))      # the 128-channel form: one K chain
        torch.cuda.synchronize()
        assert ((out - got).abs() / (1 + got.abs())).max().item() < 1e-4


def test_winograd_f4x4_occupancy_builds_agree_bit_for_bit():
    """M3D_W44_OCC2=0 selects the one-workgroup-per-CU build of the 64-channel form: same arithmetic in the same order, so the two
    builds must produce identical bits (read once per process: a child process runs the other build)."""
    code = r'''
import ctypes, sys, torch
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from m3dssd_amd import _hip
from test_gpu_round4 import _w44_case
L = _hip.lib()
d, out, ref, keep = _w44_case(128, 128, 16, 48, 2, res=True)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_hip.check(L.m3d_wino44_conv3x3_forward_ex(ctypes.byref(d), 1, st))
torch.cuda.synchronize()
torch.save(out.cpu(), sys.argv[1])
''' % (ROOT, os.path.join(ROOT, "tests"))
    import tempfile
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for occ in ("0", "1"):
            path = os.path.join(td, "o%s.pt" % occ)
            env = dict(os.environ, M3D_W44_OCC2=occ, M3D_W44_KPAIR_MAX="0")      # (K-pair workgroups add the two K halves: not this test)
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(torch.load(path))
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("nbytes", [16, 4096 + 7, 223200, 1 << 20])
def test_upload_indirect_copies_what_the_slot_names(nbytes):
    """m3d_upload_indirect: the kernel reads the source ADDRESS from an 8-byte word in pinned host memory when it runs; sizes
    that are not multiples of 16 bytes, a NULL slot (no copy) and a device-resident source."""
    L = _hip.lib()
    dev = _dev()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.RandomState(nbytes % 97)
    a = torch.from_numpy(rng.randint(0, 256, size=nbytes).astype(np.uint8)).pin_memory()
    b = torch.from_numpy(rng.randint(0, 256, size=nbytes).astype(np.uint8)).pin_memory()
    slot = torch.zeros(1, dtype=torch.int64).pin_memory()
    dst = torch.zeros(-(-nbytes // 16) * 16 + 16, dtype=torch.uint8, device=dev)
    for src in (a, b):
        slot[0] = src.data_ptr()
        _hip.check(L.m3d_upload_indirect(ctypes.c_void_p(slot.data_ptr()), ctypes.c_void_p(dst.data_ptr()), nbytes, st))
        torch.cuda.synchronize()
        assert torch.equal(dst[:nbytes].cpu(), src) and int(dst[nbytes:].sum()) == 0       # nothing past the end is touched
    slot[0] = 0
    dst.fill_(7)
    _hip.check(L.m3d_upload_indirect(ctypes.c_void_p(slot.data_ptr()), ctypes.c_void_p(dst.data_ptr()), nbytes, st))
    torch.cuda.synchronize()
    assert int((dst != 7).sum()) == 0                                                      # NULL slot: no copy
    d_src = a.to(dev)
    slot[0] = d_src.data_ptr()
    _hip.check(L.m3d_upload_indirect(ctypes.c_void_p(slot.data_ptr()), ctypes.c_void_p(dst.data_ptr()), nbytes, st))
    torch.cuda.synchronize()
    assert torch.equal(dst[:nbytes].cpu(), a)
