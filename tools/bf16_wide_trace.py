"""In-kernel timeline of bf16_conv3x3_wide_kernel (diagnostic build `make -C m3dssd_amd/csrc trace`, -DBF16_TRACE):
    python tools/bf16_wide_trace.py [cin cout H W B]
lane 0 of every wave stamps s_memtime: start | prologue done | per chunk: after positions 0, 6, 7, 12, 17 | epilogue LDS tile
written | stores issued."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from m3dssd_amd import _hip                               # noqa: E402
from m3dssd_amd.engine_bf16 import pack_conv_bf16          # noqa: E402

cin, cout, H, W, B = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (128, 128, 48, 160, 64)
dev = torch.device("cuda:0")
L = ctypes.CDLL("m3dssd_amd/csrc/build/libm3dssd_hip_trace.so")
L.m3d_conv_bf16_forward.argtypes = [ctypes.POINTER(_hip.ConvBf16Desc), ctypes.c_void_p]
L.m3d_bf16_conv_set_trace.argtypes = [ctypes.c_void_p]
g = torch.Generator().manual_seed(1)
x = torch.randn(B * H * W, cin, generator=g).to(torch.bfloat16).to(dev)
wp, kpad = pack_conv_bf16(torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5, None, None, dev)
wv = wp[:, :9 * cin].reshape(wp.shape[0] // 128, 4, 32, 9, cin // 32, 2, 2, 8).permute(0, 4, 3, 5, 1, 6, 2, 7).contiguous()
out = torch.zeros(B * H * W, cout, device=dev, dtype=torch.bfloat16)
d = _hip.ConvBf16Desc()
d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.data_ptr(), cin, B, H, W, cin
d.wgt, d.Cout, d.Cout_pad, d.Kpad = wp.data_ptr(), cout, wp.shape[0], kpad
d.kh = d.kw = 3
d.stride, d.pad, d.Ho, d.Wo = 1, 1, H, W
d.out, d.out_cs, d.out_mode, d.act, d.sigmoid_from, d.groups = out.data_ptr(), cout, 0, 1, -1, 1
d.wgt_wave = wv.data_ptr()
nw = B * (H // 8) * (W // 16)
groups = wp.shape[0] // 128
nslots = ((nw + 3) // 4) * 4 * groups
trace = torch.zeros(nslots * 64, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    assert L.m3d_conv_bf16_forward(ctypes.byref(d), st) == 0
torch.cuda.synchronize()
L.m3d_bf16_conv_set_trace(trace.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
assert L.m3d_conv_bf16_forward(ctypes.byref(d), st) == 0
e1.record()
torch.cuda.synchronize()
L.m3d_bf16_conv_set_trace(None)
t = trace.cpu().numpy().reshape(-1, 64)
t = t[t[:, 1] > 0]
NCH = cin // 32
med = lambda v: int(np.median(v))
print("launch %.4f ms; %d waves traced; %d chunks" % (e0.elapsed_time(e1), len(t), NCH))
end = 2 + 5 * NCH + 1
print("wave lifetime: median %d (min %d max %d)" % (med(t[:, end] - t[:, 0]), (t[:, end] - t[:, 0]).min(), (t[:, end] - t[:, 0]).max()))
print("prologue %d" % med(t[:, 1] - t[:, 0]))
c = t[:, 1:2 + 5 * NCH]
for k in range(NCH):
    s = c[:, 5 * k:5 * k + 6]
    print("chunk %d: pos 0 %5d | 1-6 %5d | 7 %5d | 8-12 %5d | 13-17 %5d | total %5d"
          % (k, med(s[:, 1] - s[:, 0]), med(s[:, 2] - s[:, 1]), med(s[:, 3] - s[:, 2]), med(s[:, 4] - s[:, 3]), med(s[:, 5] - s[:, 4]),
             med(s[:, 5] - s[:, 0])))
print("epilogue: math + LDS tile %d | stores %d" % (med(t[:, end - 1] - t[:, end - 2]), med(t[:, end] - t[:, end - 1])))
first = t[:, 0].min()
print("wave starts: median %d max %d after the first; ends: median %d max %d" % (med(t[:, 0] - first), (t[:, 0] - first).max(),
                                                                              med(t[:, end] - first), (t[:, end] - first).max()))
