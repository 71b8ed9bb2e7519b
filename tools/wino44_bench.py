"""Single-layer timing of the fp32 3x3 stride-1 convolutions at bs 8: Winograd F(4x4,3x3) (csrc/wino44_conv.hip) against the
F(2x2,3x3) wave kernel, with the error of both against a float64 reference.   python tools/wino44_bench.py"""
import ctypes
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from m3dssd_amd import _hip                                   # noqa: E402
from m3dssd_amd.engine import pack_wino, pack_wino44          # noqa: E402

dev = torch.device("cuda:0")
L = _hip.lib()
st = torch.cuda.current_stream().cuda_stream
COLD = "--cold" in sys.argv
flush = torch.empty(160 << 20, device=dev) if COLD else None
SHAPES = [(128, 128, 48, 160, 8), (128, 256, 48, 160, 8), (256, 256, 24, 80, 8), (64, 64, 96, 320, 8), (512, 512, 12, 40, 8)]
for cin, cout, H, W, B in SHAPES:
    g = torch.Generator().manual_seed(cin + H)
    xf = torch.randn(B, cin, H, W, generator=g)
    wf = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    x = xf.permute(0, 2, 3, 1).contiguous().to(dev)
    ref = F.conv2d(xf[:1].double(), wf.double(), padding=1).float().permute(0, 2, 3, 1)
    fl = 2.0 * B * H * W * cout * 9 * cin
    line = "%3d->%3d %3dx%3d bs%d:" % (cin, cout, H, W, B)
    kinds = ("F(2x2)", "F(4x4)", "F(4x4)+splitk")
    if "--nb" in sys.argv:                      # both workgroup forms of the F(4x4) kernel (M3D_W44_OCC2 picks the 64-channel build)
        kinds = ("F(4x4)nb1", "F(4x4)nb2")
    for kind in kinds:
        U = (pack_wino if kind == "F(2x2)" else pack_wino44)(wf, cout, dev)
        out = torch.zeros(B, H, W, cout, device=dev)
        d = _hip.ConvDesc()
        d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.data_ptr(), cin, B, H, W, cin
        d.wgt, d.Cout, d.Cout_pad = U.data_ptr(), cout, cout
        d.kh = d.kw = 3
        d.stride, d.pad, d.dil, d.Ho, d.Wo = 1, 1, 1, H, W
        d.out, d.out_cs, d.act, d.sigmoid_from = out.data_ptr(), cout, 0, -1
        if kind == "F(4x4)+splitk":
            sp, wb = ctypes.c_int(), ctypes.c_longlong()
            _hip.check(L.m3d_wino44_splitk_plan(ctypes.byref(d), ctypes.byref(sp), ctypes.byref(wb)))
            if sp.value <= 1:
                continue
            ws = torch.empty(wb.value // 4, device=dev)
            d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), wb.value
            kind += "%d" % sp.value
        fn = (lambda: L.m3d_wino_conv3x3_forward_ex(ctypes.byref(d), 1, st)) if kind == "F(2x2)" else \
            (lambda: L.m3d_wino44_conv3x3_forward(ctypes.byref(d), st))
        if kind.startswith("F(4x4)nb"):
            if kind.endswith("2") and cout % 128:
                continue
            fn = lambda nbv=int(kind[-1]): L.m3d_wino44_conv3x3_forward_ex(ctypes.byref(d), nbv, st)    # noqa: E731
        if kind == "F(2x2)":                                 # the wave kernel's own split-K form where it plans one
            sp, wb = ctypes.c_int(), ctypes.c_longlong()
            _hip.check(L.m3d_wino_conv3x3_splitk_plan(ctypes.byref(d), ctypes.byref(sp), ctypes.byref(wb)))
            if sp.value > 1:
                ws2 = torch.empty(wb.value // 4, device=dev)
                d.splitk_ws, d.splitk_ws_bytes = ws2.data_ptr(), wb.value
                fn = lambda: L.m3d_wino_conv3x3_forward(ctypes.byref(d), st)    # noqa: E731
        for _ in range(3):
            _hip.check(fn())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        if COLD:                                              # operands out of L2 / MALL: 640 MB written between launches
            ts = []
            for _ in range(7):
                flush.fill_(1.0)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            cold = sorted(ts)[len(ts) // 2]
        err = ((out[:1].cpu() - ref).abs() / (1 + ref.abs())).max().item()
        line += "  %s %.4f ms (%.0f direct-equivalent TFLOP/s, err %.1e)" % (kind, ms, fl / ms / 1e9, err)
        if COLD:
            line += " cold %.4f" % cold
    print(line, flush=True)
