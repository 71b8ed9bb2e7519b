#!/usr/bin/env python
"""Timeline of the Winograd F(2x2,3x3) bf16 kernel from the diagnostic build (make -C m3dssd_amd/csrc trace): per wave, cycles between
the stamps start | prologue staged | barrier | V(0) ready | per chunk: MFMAs + transform done, barrier passed | loop end | stores issued.
    python tools/wino2_trace.py [cin cout H W]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from m3dssd_amd import _hip  # noqa: E402
from m3dssd_amd.engine_bf16 import pack_wino2  # noqa: E402

cin, cout, H, W = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else (128, 128, 48, 160)
n = 64
dev = torch.device("cuda:0")
L = _hip.lib()
T = ctypes.CDLL(os.path.join(_hip.CSRC, "build", os.environ.get("W2_LIB", "libm3dssd_hip_trace.so")))
T.m3d_wino2_bf16_forward.argtypes = L.m3d_wino2_bf16_forward.argtypes
T.m3d_wino2_set_trace.argtypes = [ctypes.c_void_p]
g = torch.Generator().manual_seed(1)
x = torch.randn(n, H, W, cin, generator=g).to(dev, torch.bfloat16)
wf = pack_wino2(torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5, torch.ones(cout), dev)
sh = torch.zeros(cout, device=dev)
out = torch.empty(n, H, W, cout, device=dev, dtype=torch.bfloat16)
d = _hip.Wino2Bf16Desc()
d.inp, d.in_cs, d.N, d.H, d.W, d.Cin, d.Cout = x.data_ptr(), cin, n, H, W, cin, cout
d.wfrag, d.shift, d.out, d.out_cs, d.act = wf.data_ptr(), sh.data_ptr(), out.data_ptr(), cout, 1
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
nwg = n * (H // 8) * (W // 16) * (cout // 128)
trace = torch.zeros(nwg * 4 * 16, dtype=torch.int64, device=dev)
assert T.m3d_wino2_bf16_forward(ctypes.byref(d), st) == 0
torch.cuda.synchronize()
T.m3d_wino2_set_trace(trace.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
assert T.m3d_wino2_bf16_forward(ctypes.byref(d), st) == 0
e1.record()
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(nwg, 4, 16)
print("%d -> %d @ %dx%d bs %d: %d workgroups, %.3f ms (trace build)" % (cin, cout, H, W, n, nwg, e0.elapsed_time(e1)))
names = ["prologue loads + raw staged", "barrier", "V(0) + barrier"] + [w for c in range(4) for w in ("chunk %d MFMA + transform" % c, "chunk %d barrier" % c)]
dt = np.diff(t[:, :, :12], axis=2).astype(np.float64)                      # [nwg, 4, 11]
nch = min(cin // 32, 4)
for i, nm in enumerate(names[:3 + 2 * nch]):
    v = dt[:, :, i].reshape(-1)
    print("  %-32s median %7.0f  p10 %7.0f  p90 %7.0f cycles" % (nm, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
tot = (t[:, :, 13] - t[:, :, 0]).reshape(-1)
ep = (t[:, :, 13] - t[:, :, 12]).reshape(-1)
loop = (t[:, :, 12] - t[:, :, 3]).reshape(-1)
print("  K loop (all %d chunks) median %.0f, epilogue median %.0f, wave total median %.0f p90 %.0f cycles" % (cin // 32, np.median(loop), np.median(ep), np.median(tot), np.percentile(tot, 90)))
span = t[:, :, 13].max() - t[:, :, 0].min()
print("  launch span %.0f cycles of the 100 MHz-class counter units reported by s_memtime (same unit as above)" % span)
