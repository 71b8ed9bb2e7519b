"""Single-layer timing of the bf16 DCNv2 3x3 kernels at bs 64: the LDS-patch kernel (csrc/bf16_dcn_patch.hip, incl. its |offset|
pre-pass and the gated fallback launch) against the implicit-GEMM kernel, for several offset magnitudes.
    python tools/bf16_dcn_bench.py [offset_std ...]
Offsets are N(0, std) clamped to +-clamp (the window radius of the launch = ceil(max |offset|))."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from m3dssd_amd import _hip                               # noqa: E402
from m3dssd_amd.engine_bf16 import pack_conv_bf16          # noqa: E402

dev = torch.device("cuda:0")
L = _hip.lib()
SHAPES = [(128, 128, 48, 160, 64), (256, 128, 24, 80, 64), (256, 256, 24, 80, 64)]
CASES = [(0.25, 0.9), (1.0, 2.9), (1.5, 4.9), (2.0, 6.9), (3.0, 8.9)] if len(sys.argv) < 2 else \
    [(float(s), float(s) * 3.3) for s in sys.argv[1:]]
st = torch.cuda.current_stream().cuda_stream
for cin, cout, H, W, B in SHAPES:
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(B * H * W, cin, generator=g).to(torch.bfloat16).to(dev)
    wp, kpad = pack_conv_bf16(torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5, None, None, dev)
    w16 = wp.float().to(torch.float16).contiguous()
    ws = torch.zeros(max(256, L.m3d_conv_bf16_dcn_ws_bytes(B, H, W) // 4), device=dev, dtype=torch.int32)   # one flag word per pixel tile
    fl = 2.0 * B * H * W * cout * 9 * cin
    for std, clamp in CASES:
        om = torch.cat([(torch.randn(B * H * W, 18, generator=g) * std).clamp(-clamp, clamp), torch.rand(B * H * W, 9, generator=g),
                        torch.zeros(B * H * W, 5)], 1).contiguous().to(dev)
        res = {}
        for patch in (0, 1):
            out = torch.zeros(B * H * W, cout, device=dev, dtype=torch.bfloat16)
            d = _hip.ConvBf16Desc()
            d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.data_ptr(), cin, B, H, W, cin
            d.wgt, d.Cout, d.Cout_pad, d.Kpad = wp.data_ptr(), cout, wp.shape[0], kpad
            d.kh = d.kw = 3
            d.stride, d.pad, d.Ho, d.Wo = 1, 1, H, W
            d.out, d.out_cs, d.out_mode, d.act, d.sigmoid_from, d.groups = out.data_ptr(), cout, 0, 1, -1, 1
            d.dcn_offmask, d.dcn_om_cs = om.data_ptr(), 32
            if patch:
                d.wgt_f16, d.dcn_ws, d.dcn_ws_bytes = w16.data_ptr(), ws.data_ptr(), ws.numel() * 4
            var = L.m3d_conv_bf16_variant(ctypes.byref(d))
            for _ in range(3):
                _hip.check(L.m3d_conv_bf16_forward(ctypes.byref(d), st))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                L.m3d_conv_bf16_forward(ctypes.byref(d), st)
            e1.record()
            torch.cuda.synchronize()
            res[patch] = (e0.elapsed_time(e1) / 10, out.float(), var)
        diff = (res[0][1] - res[1][1]).abs().max().item() / (res[0][1].abs().max().item() + 1e-9)
        print("%3d->%3d %2dx%3d bs%d  offsets std %.2f clamp %.1f:  implicit-GEMM %.4f ms %6.1f TFLOP/s | patch(variant %d) %.4f ms %6.1f TFLOP/s"
              "  | max rel diff %.4f" % (cin, cout, H, W, B, std, clamp, res[0][0], fl / res[0][0] / 1e9, res[1][2], res[1][0],
                                        fl / res[1][0] / 1e9, diff), flush=True)
