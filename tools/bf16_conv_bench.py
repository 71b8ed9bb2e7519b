"""Single-launch timing + check of m3d_conv_bf16_forward on the 3x3 stride-1 shapes of the bf16 plan (bs 64):
    M3D_BF16_HALO={0,1,2} python tools/bf16_conv_bench.py
0 = generic implicit-GEMM tile, 1 = halo tile (one patch buffer), 2 = halo tile (two patch buffers).
Checked against torch conv2d on the bf16-rounded operands (fp32 accumulate)."""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from m3dssd_amd import _hip                               # noqa: E402
from m3dssd_amd.engine_bf16 import pack_conv_bf16          # noqa: E402

dev = torch.device("cuda:0")
L = _hip.lib()
SHAPES = [(64, 64, 96, 320, 16), (128, 128, 48, 160, 64), (256, 256, 24, 80, 64), (512, 512, 12, 40, 64), (256, 256, 32, 100, 8),
          (128, 128, 47, 157, 3), (64, 128, 96, 320, 8), (512, 256, 12, 40, 64)]
st = torch.cuda.current_stream().cuda_stream
WIDE = "--wide" in sys.argv
for cin, cout, H, W, B in SHAPES:
    g = torch.Generator().manual_seed(cin + H)
    xf = torch.randn(B, cin, H, W, generator=g).to(torch.bfloat16)
    wf = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).to(torch.bfloat16)
    x = xf.permute(0, 2, 3, 1).contiguous().to(dev)
    wp, kpad = pack_conv_bf16(wf.float(), None, None, dev)
    out = torch.full((B, H, W, cout), 7.0, device=dev, dtype=torch.bfloat16)
    d = _hip.ConvBf16Desc()
    d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.data_ptr(), cin, B, H, W, cin
    d.wgt, d.Cout, d.Cout_pad, d.Kpad = wp.data_ptr(), cout, wp.shape[0], kpad
    d.kh = d.kw = 3
    d.stride, d.pad, d.Ho, d.Wo = 1, 1, H, W
    d.out, d.out_cs, d.out_mode, d.act, d.sigmoid_from, d.groups = out.data_ptr(), cout, 0, 0, -1, 1
    if WIDE and cin % 32 == 0 and wp.shape[0] % 128 == 0:      # fragment-ordered copy for the 128 x 128 wave-tile kernel
        wv = wp[:, :9 * cin].reshape(wp.shape[0] // 128, 4, 32, 9, cin // 32, 2, 2, 8).permute(0, 4, 3, 5, 1, 6, 2, 7).contiguous()
        d.wgt_wave = wv.data_ptr()
    for _ in range(3):
        assert L.m3d_conv_bf16_forward(ctypes.byref(d), st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        L.m3d_conv_bf16_forward(ctypes.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    ref = F.conv2d(xf.float().to(dev), wf.float().to(dev), padding=1).permute(0, 2, 3, 1)
    err = (out.float() - ref).abs().max().item()
    tol = ref.abs().max().item() * 2.0 ** -7
    fl = 2.0 * B * H * W * cout * 9 * cin
    print("variant %d  halo=%s  %3d->%3d %3dx%3d bs%2d  %.4f ms  %6.1f TFLOP/s  max|err| %.4f (tol %.4f) %s"
          % (L.m3d_conv_bf16_variant(ctypes.byref(d)), os.environ.get("M3D_BF16_HALO", "1"), cin, cout, H, W, B, ms, fl / ms / 1e9, err, tol, "OK" if err <= tol else "FAIL"))
