#!/usr/bin/env python
"""Time single conv shapes through the C ABI (diagnostics: tile/ablation experiments on the GPU box).
usage: python tools/igemm_probe.py   (env M3D_ABLATE / M3D_BM_THRESHOLD / M3D_FORCE_BN select variants)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from m3dssd_amd import _hip
from m3dssd_amd.host import standalone as S

SHAPES = [  # name, N, Cin, H, W, Cout, k, stride
    ("l3 3x3 128->128 @48x160", 8, 128, 48, 160, 128, 3, 1),
    ("cls.0 3x3 128->256 @48x160", 8, 128, 48, 160, 256, 3, 1),
    ("l4 3x3 256->256 @24x80", 8, 256, 24, 80, 256, 3, 1),
    ("l5 3x3 512->512 @12x40", 8, 512, 12, 40, 512, 3, 1),
    ("head.3 1x1 256->256 @48x160", 8, 256, 48, 160, 256, 1, 1),
    ("head.0 1x1 128->256 @48x160", 8, 128, 48, 160, 256, 1, 1),
    ("l2 3x3 64->64 @96x320", 8, 64, 96, 320, 64, 3, 1),
]
dev = torch.device("cuda:0")
L = _hip.lib()
for name, n, ci, h, w, co, k, stride in SHAPES:
    x = torch.randn(n, h, w, ci, device=dev)
    v = S.View(x, n, h, w, ci, ci)
    wt = torch.randn(co, ci, k, k, device=dev) / (ci * k * k) ** 0.5
    wp, co_, cop, kh, kw = S._pack(wt, ci, 32)
    use_wino = os.environ.get("PROBE_WINO", "0") == "1" and k == 3 and stride == 1
    if use_wino:
        from m3dssd_amd.engine import pack_wino
        wp = pack_wino(wt, cop, dev)
    fn = L.m3d_wino_conv3x3_forward if use_wino else L.m3d_conv2d_forward
    ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
    out = torch.empty(n, ho, wo, co, device=dev)
    d = _hip.ConvDesc()
    d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = v.ptr, ci, n, h, w, ci
    d.wgt, d.Cout, d.Cout_pad = wp.data_ptr(), co, cop
    d.kh, d.kw, d.stride, d.pad, d.dil, d.Ho, d.Wo = k, k, stride, k // 2, 1, ho, wo
    d.out, d.out_cs, d.act, d.sigmoid_from = out.data_ptr(), co, 1, -1
    st = S._stream()
    bm, bn, bk, grid = (ctypes.c_int() for _ in range(4))
    L.m3d_conv2d_tile(ctypes.byref(d), ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(bk), ctypes.byref(grid))
    for _ in range(3):
        _hip.check(fn(ctypes.byref(d), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20
    e0.record()
    for _ in range(iters):
        _hip.check(fn(ctypes.byref(d), st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    gf = 2.0 * n * ho * wo * co * k * k * ci / 1e9
    print("%-30s tile %dx%dx%d grid %5d  %7.3f ms  %6.1f TF" % (name, bm.value, bn.value, bk.value, grid.value, ms, gf / ms))
