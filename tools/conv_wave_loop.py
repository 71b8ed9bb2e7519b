"""Runs one kernel family back to back for ~8 s (clock / power sampling with tools/clock_probe.sh):
python tools/conv_wave_loop.py deform|plain|wino|mfma"""
import ctypes
import subprocess
import sys
import time

import torch

sys.path.insert(0, ".")
from m3dssd_amd import _hip                       # noqa: E402
from m3dssd_amd.engine import pack_frag, pack_wino           # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "deform"
if mode == "mfma":
    subprocess.run(["tools/ubench/mfma_loop.bin"])
    sys.exit(0)
dev = torch.device("cuda:0")
L = _hip.lib()
import os
if os.environ.get('TRACE_LIB'):
    L = ctypes.CDLL('m3dssd_amd/csrc/build/libm3dssd_hip_trace.so')
    for f in ('m3d_wino_conv3x3_forward', 'm3d_conv_wave_forward'):
        getattr(L, f).argtypes = [ctypes.POINTER(_hip.ConvDesc), ctypes.c_void_p]
B, H, W, cin, cout = 8, 48, 160, 128, 128
x = torch.randn(B * H * W * cin, device=dev)
out = torch.empty(B * H * W * cout, device=dev)
d = _hip.ConvDesc()
d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.data_ptr(), cin, B, H, W, cin
d.Cout, d.Cout_pad = cout, cout
d.kh = d.kw = 3
d.stride, d.pad, d.dil, d.Ho, d.Wo = 1, 1, 1, H, W
d.out, d.out_cs, d.act, d.sigmoid_from = out.data_ptr(), cout, 1, -1
if mode == "wino":
    wf = pack_wino(torch.randn(cout, cin, 3, 3) / 34.0, cout, dev)
    fn = lambda st: L.m3d_wino_conv3x3_forward(ctypes.byref(d), st)
else:
    wf = pack_frag(torch.randn(cout, 9 * cin) / 34.0, cout, dev)
    fn = lambda st: L.m3d_conv_wave_forward(ctypes.byref(d), st)
    if mode == "deform":
        om = torch.cat([torch.randn(B * H * W, 18, device=dev), torch.rand(B * H * W, 9, device=dev), torch.zeros(B * H * W, 1, device=dev)], 1).contiguous()
        d.dcn_offmask, d.dcn_om_cs = om.data_ptr(), 28
d.wgt = wf.data_ptr()
st = torch.cuda.current_stream().cuda_stream
t0 = time.time()
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
while time.time() - t0 < float(os.environ.get('LOOP_S', '8')):
    e0.record()
    for _ in range(200):
        assert fn(st) == 0
    e1.record()
    torch.cuda.synchronize()
    n += 200
    ms = e0.elapsed_time(e1) / 200
fl = 2.0 * B * H * W * cout * 9 * cin / (2.25 if mode == "wino" else 1.0)
print("%s: %d launches, last batch %.4f ms/launch = %.1f executed TFLOP/s" % (mode, n, ms, fl / ms / 1e9))
