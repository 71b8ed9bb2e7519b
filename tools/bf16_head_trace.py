"""In-kernel timeline of bf16_head_mlp_kernel (diagnostic build `make -C m3dssd_amd/csrc trace`):
    python tools/bf16_head_trace.py [groups] [B]
Stamps of thread 0 of every workgroup: start | input tile staged | after each of the 10 weight-chunk steps and the two
hidden-tile writes (every barrier) | MFMAs of the last chunk done | outputs stored."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from m3dssd_amd import _hip                               # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
HW = 48 * 160
M = B * HW
dev = torch.device("cuda:0")
L = ctypes.CDLL("m3dssd_amd/csrc/build/libm3dssd_hip_trace.so")
L.m3d_head_mlp_bf16_forward.argtypes = [ctypes.POINTER(_hip.HeadBf16Desc), ctypes.c_void_p]
L.m3d_bf16_head_set_trace.argtypes = [ctypes.c_void_p]
bf = torch.bfloat16
x = torch.randn(M, 128, device=dev).to(bf)
w1 = (torch.randn(G, 256, 128, device=dev) / 11).to(bf)
w2 = (torch.randn(G, 256, 256, device=dev) / 16).to(bf)
w3 = (torch.randn(G, 64, 256, device=dev) / 16).to(bf)
s = [torch.ones(G, 256, device=dev), torch.zeros(G, 256, device=dev), torch.ones(G, 256, device=dev), torch.zeros(G, 256, device=dev),
     torch.ones(G, 36, device=dev), torch.zeros(G, 36, device=dev)]
out = torch.empty(G, B, 36, HW, device=dev)
d = _hip.HeadBf16Desc()
d.inp, d.in_cs, d.M, d.Cin = x.data_ptr(), 128, M, 128
d.w1, d.w2, d.w3 = w1.data_ptr(), w2.data_ptr(), w3.data_ptr()
d.s1, d.t1, d.s2, d.t2, d.s3, d.t3 = (t.data_ptr() for t in s)
d.Cout, d.Cout_pad, d.out, d.out_group_off, d.out_img_stride, d.HW, d.groups = 36, 64, out.data_ptr(), B * 36 * HW, 36 * HW, HW, G
st = torch.cuda.current_stream().cuda_stream
nblk = min((M + 127) // 128, max(1, 256 // G)) * G          # one workgroup per CU, each walks tiles
trace = torch.zeros(nblk * 32, dtype=torch.int64, device=dev)
for _ in range(3):
    assert L.m3d_head_mlp_bf16_forward(ctypes.byref(d), st) == 0
torch.cuda.synchronize()
L.m3d_bf16_head_set_trace(trace.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
assert L.m3d_head_mlp_bf16_forward(ctypes.byref(d), st) == 0
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
fl = 2.0 * M * G * (128 * 256 + 256 * 256 + 256 * 64)
t = trace.cpu().numpy().reshape(nblk, 32)
n = int((t[0] != 0).sum())
print("%d workgroups, launch %.3f ms (%.0f TFLOP/s executed), %d stamps" % (nblk, ms, fl / ms / 1e9, n))
dt = np.diff(t[:, :n], axis=1)
names = ["affine -> LDS, first input / weight loads, stage", "L1 chunk0", "L1 chunk1 + hidden1 write", "L2 chunk0", "L2 chunk1",
         "L2 chunk2", "L2 chunk3", "hidden2 write", "L3 chunk0", "L3 chunk1", "L3 chunk2", "L3 chunk3 MFMAs", "barrier",
         "output: LDS transpose + 16-byte stores", "stage next input + chunk 0"]
for i in range(n - 1):
    nm = names[i] if i < len(names) else names[(i - len(names)) % (len(names) - 1) + 1] + " (tile 2)"
    print("  %-52s median %6d  p90 %6d" % (nm, int(np.median(dt[:, i])), int(np.percentile(dt[:, i], 90))))
per_tile = t[:, 15] - t[:, 1] if n > 15 else None
if per_tile is not None:
    print("  one tile (stamp 1 -> 15) median %d cycles" % int(np.median(per_tile)))
