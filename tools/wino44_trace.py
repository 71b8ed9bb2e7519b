"""In-kernel timeline of wino44_kernel (diagnostic build `make -C m3dssd_amd/csrc trace`, -DWINO_TRACE):
    python tools/wino44_trace.py [Cin] [Cout] [H] [W] [B] [nb]
lane 0 of waves 0 and 3 of every workgroup stamps s_memtime: start | first patch landed | transformed + stored | barrier |
per stage: MFMA loop done, next transform stored, barrier passed | epilogue done."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from m3dssd_amd import _hip                                   # noqa: E402
from m3dssd_amd.engine import pack_wino44                     # noqa: E402

cin = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 128
H = int(sys.argv[3]) if len(sys.argv) > 3 else 48
W = int(sys.argv[4]) if len(sys.argv) > 4 else 160
B = int(sys.argv[5]) if len(sys.argv) > 5 else 8
nb = int(sys.argv[6]) if len(sys.argv) > 6 else 0
dev = torch.device("cuda:0")
L = ctypes.CDLL("m3dssd_amd/csrc/build/libm3dssd_hip_trace.so")
L.m3d_wino44_conv3x3_forward_ex.argtypes = [ctypes.POINTER(_hip.ConvDesc), ctypes.c_int, ctypes.c_void_p]
L.m3d_wino44_set_trace.argtypes = [ctypes.c_void_p]
x = torch.randn(B * H * W * cin, device=dev)
U = pack_wino44(torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5, cout, dev)
out = torch.empty(B * H * W * cout, device=dev)
d = _hip.ConvDesc()
d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.data_ptr(), cin, B, H, W, cin
d.wgt, d.Cout, d.Cout_pad = U.data_ptr(), cout, cout
d.kh = d.kw = 3
d.stride, d.pad, d.dil, d.Ho, d.Wo = 1, 1, 1, H, W
d.out, d.out_cs, d.act, d.sigmoid_from = out.data_ptr(), cout, 1, -1
strips = -(-(B * H * W // 16) // 16)
grid = strips * (cout // 64)
trace = torch.zeros(grid * 2 * 64, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    assert L.m3d_wino44_conv3x3_forward_ex(ctypes.byref(d), nb, st) == 0
torch.cuda.synchronize()
L.m3d_wino44_set_trace(trace.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
assert L.m3d_wino44_conv3x3_forward_ex(ctypes.byref(d), nb, st) == 0
e1.record()
torch.cuda.synchronize()
L.m3d_wino44_set_trace(None)
t = trace.cpu().numpy().reshape(grid, 2, 64)
t = t[t[:, 0, 0] > 0]
NS = cin // 16
med = lambda v: int(np.median(v))
print("launch %.4f ms, %d workgroups traced, %d stages" % (e0.elapsed_time(e1), len(t), NS))
for w, name in ((0, "wave 0"), (1, "wave 3")):
    q = t[:, w]
    print("%s: first patch landed %d | transform + store %d | first fragments / patch issued + barrier %d" %
          (name, med(q[:, 1] - q[:, 0]), med(q[:, 2] - q[:, 1]), med(q[:, 3] - q[:, 2])))
    i = 3
    for s in range(NS - 1):
        print("   stage %2d: 288 (or 144) MFMAs %5d | transform + store + next patch issued %5d | barrier %5d" %
              (s, med(q[:, i + 1] - q[:, i]), med(q[:, i + 2] - q[:, i + 1]), med(q[:, i + 3] - q[:, i + 2])))
        i += 3
    print("   last stage: MFMAs %5d | A^T M A + epilogue %5d | lifetime %d" % (med(q[:, i + 1] - q[:, i]), med(q[:, i + 2] - q[:, i + 1]), med(q[:, i + 2] - q[:, 0])))
