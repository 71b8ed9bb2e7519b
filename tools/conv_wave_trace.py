"""In-kernel timeline of conv_wave_kernel (diagnostic build -DCONV_TRACE): python tools/conv_wave_trace.py [deform] [Cin] [Cout] [H] [W] [B]"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from m3dssd_amd import _hip                       # noqa: E402
from m3dssd_amd.engine import pack_frag           # noqa: E402

deform = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 128
H = int(sys.argv[4]) if len(sys.argv) > 4 else 48
W = int(sys.argv[5]) if len(sys.argv) > 5 else 160
B = int(sys.argv[6]) if len(sys.argv) > 6 else 8
k = int(sys.argv[7]) if len(sys.argv) > 7 else 3
dev = torch.device("cuda:0")
L = ctypes.CDLL("m3dssd_amd/csrc/build/libm3dssd_hip_trace.so")
L.m3d_conv_wave_forward.argtypes = [ctypes.POINTER(_hip.ConvDesc), ctypes.c_void_p]
L.m3d_conv_wave_set_trace.argtypes = [ctypes.c_void_p]
x = torch.randn(B * H * W * cin, device=dev)
wf = pack_frag(torch.randn(cout, k * k * cin) / (k * k * cin) ** 0.5, cout, dev)
om = torch.cat([torch.randn(B * H * W, 18, device=dev), torch.rand(B * H * W, 9, device=dev), torch.zeros(B * H * W, 1, device=dev)], 1).contiguous()
out = torch.empty(B * H * W * cout, device=dev)
d = _hip.ConvDesc()
d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.data_ptr(), cin, B, H, W, cin
d.wgt, d.Cout, d.Cout_pad = wf.data_ptr(), cout, cout
d.kh = d.kw = k
d.stride, d.pad, d.dil, d.Ho, d.Wo = 1, k // 2, 1, H, W
d.out, d.out_cs, d.act, d.sigmoid_from = out.data_ptr(), cout, 1, -1
if deform:
    d.dcn_offmask, d.dcn_om_cs = om.data_ptr(), 28
grid = (B * H * W + 31) // 32 * (cout // 128)
trace = torch.zeros(grid * 128, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    assert L.m3d_conv_wave_forward(ctypes.byref(d), st) == 0
torch.cuda.synchronize()
L.m3d_conv_wave_set_trace(trace.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
assert L.m3d_conv_wave_forward(ctypes.byref(d), st) == 0
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
t = trace.cpu().numpy().reshape(grid, 128)
dur = t.max(axis=1) - t[:, 0]
fl = 2.0 * B * H * W * cout * k * k * cin
print("grid %d waves, launch %.4f ms (%.1f TFLOP/s); wave durations: min %d median %d max %d ticks" % (grid, ms, fl / ms / 1e9, dur.min(), int(np.median(dur)), dur.max()))
for blk in (grid // 2, grid // 2 + 7):
    s = t[blk]
    s = s[s > 0] - t[blk, 0]
    dl = np.diff(s)
    ns = min(12, (len(dl) - 2) // 2)
    print("wave %d: prologue %d | per step (combine+issue, mfma) first %d: %s | ... last: %s" % (
        blk, s[1], ns, " ".join("%d,%d" % (dl[1 + 2 * i], dl[2 + 2 * i]) for i in range(ns)), " ".join("%d" % v for v in dl[-4:])))
ep = np.array([tt[tt > 0][-1] - tt[tt > 0][-2] for tt in t])
print("epilogue (last stamp pair): median %d ticks = %.0f %% of the wave" % (int(np.median(ep)), 100.0 * np.median(ep) / np.median(dur)))
