"""In-kernel timeline of the Winograd kernel (diagnostic build -DWINO_TRACE, see csrc/wino_conv.hip):
python tools/wino_trace.py [Cin] [H] [W] [B]   -> per-wave s_memtime stamps of a first-round and a last-round block."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from m3dssd_amd import _hip                       # noqa: E402
from m3dssd_amd.engine import pack_wino           # noqa: E402

cin = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 24
W = int(sys.argv[3]) if len(sys.argv) > 3 else 80
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
cout = cin
dev = torch.device("cuda:0")
L = ctypes.CDLL("m3dssd_amd/csrc/build/libm3dssd_hip_trace.so")
L.m3d_wino_conv3x3_forward.argtypes = [ctypes.POINTER(_hip.ConvDesc), ctypes.c_void_p]
L.m3d_wino_set_trace.argtypes = [ctypes.c_void_p]
x = torch.randn(B * H * W * cin, device=dev)
w = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5
U = pack_wino(w, cout, dev)
out = torch.empty(B * H * W * cout, device=dev)
d = _hip.ConvDesc()
d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.data_ptr(), cin, B, H, W, cin
d.wgt, d.Cout, d.Cout_pad = U.data_ptr(), cout, cout
d.kh = d.kw = 3
d.stride, d.pad, d.dil, d.Ho, d.Wo = 1, 1, 1, H, W
d.out, d.out_cs, d.act, d.sigmoid_from = out.data_ptr(), cout, 1, -1
if os.environ.get("RES"):
    d.res, d.res_cs = x.data_ptr(), cin
variant = int(os.environ.get('M3D_WINO_VARIANT', '1'))
grid = -(-(B * H * W // 4) // (32 if variant == 1 else 64)) * (cout // 32)
trace = torch.zeros(grid * 8 * 128, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    assert L.m3d_wino_conv3x3_forward(ctypes.byref(d), st) == 0
torch.cuda.synchronize()
L.m3d_wino_set_trace(trace.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
assert L.m3d_wino_conv3x3_forward(ctypes.byref(d), st) == 0
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
t = trace.cpu().numpy().reshape(grid, 8, 128)
start = t[:, :, 0][t[:, :, 0] > 0].min()
end = t.max()
print("grid %d blocks, launch %.4f ms, stamps span %d ticks (%.1f ticks/us)" % (grid, ms, end - start, (end - start) / (ms * 1e3)))
rt = t[:, 0, 127] - t[:, 0, 126]
last = np.where(t[:, 0, :126] > 0, t[:, 0, :126], 0).max(axis=1)
clk = (last - t[:, 0, 0]) / np.maximum(rt, 1) * 0.1
print("shader clock inside the kernel (memtime/realtime per wave 0): median %.3f GHz, min %.3f, max %.3f" % (np.median(clk), clk.min(), clk.max()))
t = t.copy(); t[:, :, 126:] = 0
bstart = t[:, 0, 0] - start
order = np.argsort(bstart)
print("block start times (ticks): first 5", bstart[order[:5]], " median", int(np.median(bstart)), " last 5", bstart[order[-5:]])
bend = t.max(axis=(1, 2)) - start
print("block durations: min %d median %d max %d" % ((bend - bstart).min(), int(np.median(bend - bstart)), (bend - bstart).max()))
for blk in (int(order[0]), int(order[-1])):
    print("---- block %d (start %d)" % (blk, bstart[blk]))
    for wv in ((0,) if variant == 1 else (0, 4)):
        s = t[blk, wv]
        s = s[s > 0] - t[blk, 0, 0]
        print("wave %d: %d stamps" % (wv, len(s)))
        print("  ", " ".join("%d" % v for v in s))
