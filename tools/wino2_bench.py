#!/usr/bin/env python
"""m3d_wino2_bf16_forward against the kernel the plan used before (m3d_conv_bf16_forward, wave-tile form) on the plan's layer shapes
at bs 64: HIP-event time per launch, algorithmic TFLOP/s, max difference between the two.  python tools/wino2_bench.py [reps]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from m3dssd_amd import _hip  # noqa: E402
from m3dssd_amd.engine_bf16 import pack_wino2  # noqa: E402

L = _hip.lib()
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
BF16 = torch.bfloat16
for (n, cin, cout, H, W, res) in [(64, 128, 128, 48, 160, False), (64, 128, 128, 48, 160, True), (64, 256, 256, 24, 80, False),
                                  (64, 256, 256, 24, 80, True), (64, 128, 256, 48, 160, False)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, H, W, cin, generator=g).to(dev, BF16)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).to(BF16).float()
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    r = torch.randn(n, H, W, cout, generator=g).to(dev, BF16) if res else None
    out = torch.empty(n, H, W, cout, device=dev, dtype=BF16)
    wf = pack_wino2(w, sc, dev)
    shd = sh.to(dev)
    d = _hip.Wino2Bf16Desc()
    d.inp, d.in_cs, d.N, d.H, d.W, d.Cin, d.Cout = x.data_ptr(), cin, n, H, W, cin, cout
    d.wfrag, d.shift, d.out, d.out_cs, d.act = wf.data_ptr(), shd.data_ptr(), out.data_ptr(), cout, 1
    if res:
        d.res, d.res_cs = r.data_ptr(), cout
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        _hip.check(L.m3d_wino2_bf16_forward(ctypes.byref(d), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _hip.check(L.m3d_wino2_bf16_forward(ctypes.byref(d), st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gf = 2.0 * n * H * W * cin * cout * 9 / 1e9
    # reference on a slice (torch conv in fp32 on the same rounded operands)
    xs = x[:2].float().permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xs, w.to(dev), None, padding=1) * sc.to(dev).view(1, -1, 1, 1) + shd.view(1, -1, 1, 1)
    if res:
        ref = ref + r[:2].float().permute(0, 3, 1, 2)
    ref = torch.nn.functional.leaky_relu(ref, 0.01)
    err = (out[:2].float().permute(0, 3, 1, 2) - ref).abs().max().item()
    print("%3d x %3d -> %3d @ %2dx%3d res %d: %.3f ms  %.0f TFLOP/s (direct-conv FLOPs)  max err %.4f of %.2f" % (
        n, cin, cout, H, W, int(res), ms, gf / ms, err, ref.abs().max().item()))
