"""In-kernel timeline of bf16_dcn_patch_kernel<16, 9, 3> (diagnostic build `make -C m3dssd_amd/csrc trace`, -DBF16_TRACE):
    python tools/bf16_dcn_trace.py [offset std] [clamp]
thread 0 of every workgroup stamps s_memtime: start | bound reduced | first loads issued | prologue staged | per weight stage
(3 taps x 32 channels): top, compute issued, weights stored, barrier passed | epilogue phases."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from m3dssd_amd import _hip                               # noqa: E402
from m3dssd_amd.engine_bf16 import pack_conv_bf16          # noqa: E402

std = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
clamp = float(sys.argv[2]) if len(sys.argv) > 2 else 4.9
cin, cout, H, W, B = 128, 128, 48, 160, 64
dev = torch.device("cuda:0")
L = ctypes.CDLL("m3dssd_amd/csrc/build/libm3dssd_hip_trace.so")
L.m3d_conv_bf16_forward.argtypes = [ctypes.POINTER(_hip.ConvBf16Desc), ctypes.c_void_p]
L.m3d_bf16_conv_set_trace.argtypes = [ctypes.c_void_p]
g = torch.Generator().manual_seed(1)
x = torch.randn(B * H * W, cin, generator=g).to(torch.bfloat16).to(dev)
wp, kpad = pack_conv_bf16(torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5, None, None, dev)
w16 = wp.float().to(torch.float16).contiguous()
ws = torch.zeros(max(256, L.m3d_conv_bf16_dcn_ws_bytes(B, H, W) // 4), device=dev, dtype=torch.int32)   # one flag word per pixel tile
om = torch.cat([(torch.randn(B * H * W, 18, generator=g) * std).clamp(-clamp, clamp), torch.rand(B * H * W, 9, generator=g),
                torch.zeros(B * H * W, 5)], 1).contiguous().to(dev)
out = torch.zeros(B * H * W, cout, device=dev, dtype=torch.bfloat16)
d = _hip.ConvBf16Desc()
d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.data_ptr(), cin, B, H, W, cin
d.wgt, d.Cout, d.Cout_pad, d.Kpad = wp.data_ptr(), cout, wp.shape[0], kpad
d.kh = d.kw = 3
d.stride, d.pad, d.Ho, d.Wo = 1, 1, H, W
d.out, d.out_cs, d.out_mode, d.act, d.sigmoid_from, d.groups = out.data_ptr(), cout, 0, 1, -1, 1
d.dcn_offmask, d.dcn_om_cs = om.data_ptr(), 32
d.wgt_f16, d.dcn_ws, d.dcn_ws_bytes = w16.data_ptr(), ws.data_ptr(), ws.numel() * 4
grid = B * (H // 16) * (W // 16)
trace = torch.zeros(max(grid, B * H * W // 128) * 160, dtype=torch.int64, device=dev)
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
t = trace.cpu().numpy().reshape(-1, 160)[:grid]
NS = 12
med = lambda v: int(np.median(v))
print("launch (bound + patch + gated fallback) %.4f ms; %d workgroups" % (e0.elapsed_time(e1), grid))
life = t[:, 8 + 4 * NS] - t[:, 0]
print("workgroup lifetime: median %d cycles (min %d max %d)" % (med(life), life.min(), life.max()))
print("prologue: om loads + bound reduce %d | first window / weight loads issued %d | sampling states %d | wait + convert + stage %d"
      % (med(t[:, 1] - t[:, 0]), med(t[:, 2] - t[:, 1]), med(t[:, 3] - t[:, 2]), med(t[:, 4] - t[:, 3])))
t = np.concatenate([t[:, :3], t[:, 4:]], 1)            # drop the extra stamp: the layout below is the original one
s = t[:, 4:4 + 4 * NS].reshape(grid, NS, 4)
top = np.concatenate([t[:, 3:4], s[:, :-1, 3]], 1)
print("per stage (3 taps x 32 ch = 24 MFMAs = 768 MFMA cycles per wave, 1536 per SIMD):")
for i in range(NS):
    print("  stage %2d: issue loads + compute %5d | weight ds_write %4d | (window restage +) barrier %5d | total %5d"
          % (i, med(s[:, i, 1] - s[:, i, 0]), med(s[:, i, 2] - s[:, i, 1]), med(s[:, i, 3] - s[:, i, 2]), med(s[:, i, 3] - top[:, i])))
b = 4 + 4 * NS
print("epilogue: affine -> LDS + barrier %d | accumulators -> LDS tile %d | barrier + stores issued %d"
      % (med(t[:, b] - s[:, -1, 3]), med(t[:, b + 1] - t[:, b]), med(t[:, b + 2] - t[:, b + 1])))
