"""In-kernel timeline of the LDS-tiled igemm (diagnostic build -DIGEMM_TRACE): python tools/igemm_trace.py [deform] [Cin] [Cout] [H] [W] [B]
Stamps per wave: start, pre-barrier, post-barrier, then per k-tile: loads issued, MFMAs issued, tile stored, barrier passed."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from m3dssd_amd import _hip                       # noqa: E402

deform = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 128
H = int(sys.argv[4]) if len(sys.argv) > 4 else 48
W = int(sys.argv[5]) if len(sys.argv) > 5 else 160
B = int(sys.argv[6]) if len(sys.argv) > 6 else 8
dev = torch.device("cuda:0")
L = ctypes.CDLL("m3dssd_amd/csrc/build/libm3dssd_hip_trace.so")
L.m3d_conv2d_forward.argtypes = [ctypes.POINTER(_hip.ConvDesc), ctypes.c_void_p]
L.m3d_conv2d_tile.argtypes = [ctypes.POINTER(_hip.ConvDesc)] + [ctypes.POINTER(ctypes.c_int)] * 4
L.m3d_igemm_set_trace.argtypes = [ctypes.c_void_p]
x = torch.randn(B * H * W * cin, device=dev)
wp = torch.randn(cout * 9 * cin, device=dev) / (9 * cin) ** 0.5
om = torch.cat([torch.randn(B * H * W, 18, device=dev), torch.rand(B * H * W, 9, device=dev), torch.zeros(B * H * W, 1, device=dev)], 1).contiguous()
out = torch.empty(B * H * W * cout, device=dev)
d = _hip.ConvDesc()
d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.data_ptr(), cin, B, H, W, cin
d.wgt, d.Cout, d.Cout_pad = wp.data_ptr(), cout, cout
d.kh = d.kw = 3
d.stride, d.pad, d.dil, d.Ho, d.Wo = 1, 1, 1, H, W
d.out, d.out_cs, d.act, d.sigmoid_from = out.data_ptr(), cout, 1, -1
if deform:
    d.dcn_offmask, d.dcn_om_cs = om.data_ptr(), 28
bm, bn, bk, grid = (ctypes.c_int() for _ in range(4))
L.m3d_conv2d_tile(ctypes.byref(d), ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(bk), ctypes.byref(grid))
grid = grid.value
trace = torch.zeros(grid * 4 * 128, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    assert L.m3d_conv2d_forward(ctypes.byref(d), st) == 0
torch.cuda.synchronize()
L.m3d_igemm_set_trace(trace.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
assert L.m3d_conv2d_forward(ctypes.byref(d), st) == 0
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
t = trace.cpu().numpy().reshape(grid, 4, 128)
print("tile %dx%dx%d grid %d blocks, launch %.4f ms" % (bm.value, bn.value, bk.value, grid, ms))
dur = t.max(axis=(1, 2)) - t[:, 0, 0]
print("block durations: min %d median %d max %d ticks" % (dur.min(), int(np.median(dur)), dur.max()))
for blk in (0, grid // 2):
    for wv in (0, 3):
        s = t[blk, wv]
        s = s[s > 0] - t[blk, 0, 0]
        print("block %d wave %d: %d stamps" % (blk, wv, len(s)))
        print("  ", " ".join("%d" % v for v in s[:3]), " | ", "  ".join(",".join("%d" % v for v in s[3 + 4 * i: 7 + 4 * i]) for i in range(min(9, (len(s) - 3) // 4))))
