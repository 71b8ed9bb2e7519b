"""Times m3d_conv_wave_forward against m3d_conv2d_forward (LDS-tiled) for 1x1 convolutions: python tools/conv1x1_probe.py"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from m3dssd_amd import _hip                       # noqa: E402
from m3dssd_amd.engine import pack_frag           # noqa: E402

dev = torch.device("cuda:0")
L = _hip.lib()
st = torch.cuda.current_stream().cuda_stream
M = 8 * 48 * 160
for cin, cout in ((128, 512), (192, 384), (384, 128), (64, 128), (256, 256)):
    x = torch.randn(M * cin, device=dev)
    w = torch.randn(cout, cin, device=dev) / cin ** 0.5
    wf = pack_frag(w, cout, dev)
    out = torch.empty(M * cout, device=dev)
    res = {}
    for name, wp, fn in (("block", w.contiguous(), L.m3d_conv2d_forward), ("wave", wf, L.m3d_conv_wave_forward)):
        d = _hip.ConvDesc()
        d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.data_ptr(), cin, 8, 48, 160, cin
        d.wgt, d.Cout, d.Cout_pad = wp.data_ptr(), cout, cout
        d.kh = d.kw = 1
        d.stride, d.pad, d.dil, d.Ho, d.Wo = 1, 0, 1, 48, 160
        d.out, d.out_cs, d.act, d.sigmoid_from = out.data_ptr(), cout, 0, -1
        for _ in range(3):
            _hip.check(fn(ctypes.byref(d), st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            _hip.check(fn(ctypes.byref(d), st))
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20
    fl = 2.0 * M * cin * cout
    print("1x1 %3d -> %3d: block %.4f ms (%.1f TF)  wave %.4f ms (%.1f TF)" % (cin, cout, res["block"], fl / res["block"] / 1e9,
                                                                          res["wave"], fl / res["wave"] / 1e9))
