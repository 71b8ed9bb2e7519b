"""Fused bf16 heads: launch times of both forms back to back, and -- with the diagnostic build `make -C m3dssd_amd/csrc trace` --
the in-kernel timeline of bf16_head2_kernel (thread 0 of every workgroup, its SECOND tile: start | input staged (barrier A) |
layer 1 + h1 written (barrier B) | layer 2 MFMAs | h2 complete (barrier D) | layer 3 MFMAs | outputs stored).
    python tools/bf16_head2_trace.py [groups] [B]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from m3dssd_amd import _hip                               # noqa: E402
from m3dssd_amd.engine_bf16 import pack_head2             # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
HW = 48 * 160
M = B * HW
dev = torch.device("cuda:0")
bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
x = torch.randn(M, 128, generator=g).to(bf).to(dev)
w1 = torch.randn(G, 256, 128, generator=g) / 11
w2 = torch.randn(G, 256, 256, generator=g) / 16
w3 = torch.randn(G, 36, 256, generator=g) / 16
s = [torch.rand(G, c, generator=g) + 0.5 for c in (256, 256, 36)]
t = [torch.randn(G, c, generator=g) * 0.1 for c in (256, 256, 36)]
out = torch.empty(G, B, 36, HW, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
L = _hip.lib()

w3p = torch.zeros(G, 64, 256)
w3p[:, :36] = w3
dv = [v.to(dev).contiguous() for v in (w1.to(bf), w2.to(bf), w3p.to(bf), s[0], t[0], s[1], t[1], s[2], t[2])]
d1 = _hip.HeadBf16Desc()
d1.inp, d1.in_cs, d1.M, d1.Cin = x.data_ptr(), 128, M, 128
d1.w1, d1.w2, d1.w3, d1.s1, d1.t1, d1.s2, d1.t2, d1.s3, d1.t3 = (v.data_ptr() for v in dv)
d1.Cout, d1.Cout_pad, d1.out, d1.out_group_off, d1.out_img_stride, d1.HW, d1.groups = 36, 64, out.data_ptr(), B * 36 * HW, 36 * HW, HW, G
pk = pack_head2([(w1[i], s[0][i], t[0][i], w2[i], s[1][i], t[1][i], w3[i], s[2][i], t[2][i]) for i in range(G)], dev)
d2 = _hip.Head2Bf16Desc()
d2.inp, d2.in_cs, d2.M = x.data_ptr(), 128, M
d2.w1f, d2.w2f, d2.w3, d2.t1, d2.t2, d2.t3 = (v.data_ptr() for v in pk)
d2.Cout, d2.out, d2.out_group_off, d2.out_img_stride, d2.HW, d2.groups = 36, out.data_ptr(), B * 36 * HW, 36 * HW, HW, G


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


fl = 2.0 * M * G * (128 * 256 + 256 * 256 + 256 * 36)
ms1 = timeit(lambda: _hip.check(L.m3d_head_mlp_bf16_forward(ctypes.byref(d1), st)))
o1 = out.clone()
ms2 = timeit(lambda: _hip.check(L.m3d_head_mlp2_bf16_forward(ctypes.byref(d2), st)))
print("G = %d, B = %d: form 1 %.4f ms (%.0f TFLOP/s), form 2 %.4f ms (%.0f TFLOP/s); max |form2 - form1| = %.3e (max |out| %.2f)"
      % (G, B, ms1, fl / ms1 / 1e9, ms2, fl / ms2 / 1e9, (out - o1).abs().max().item(), o1.abs().max().item()))
tp = "m3dssd_amd/csrc/build/libm3dssd_hip_trace.so"
if os.path.exists(tp):
    T = ctypes.CDLL(tp)
    if hasattr(T, "m3d_bf16_head2_set_trace"):
        T.m3d_head_mlp2_bf16_forward.argtypes = L.m3d_head_mlp2_bf16_forward.argtypes
        T.m3d_bf16_head2_set_trace.argtypes = [ctypes.c_void_p]
        nblk = max(1, 256 // G) * G
        trace = torch.zeros(nblk * 16, dtype=torch.int64, device=dev)
        assert T.m3d_head_mlp2_bf16_forward(ctypes.byref(d2), st) == 0
        torch.cuda.synchronize()
        T.m3d_bf16_head2_set_trace(trace.data_ptr())
        assert T.m3d_head_mlp2_bf16_forward(ctypes.byref(d2), st) == 0
        torch.cuda.synchronize()
        T.m3d_bf16_head2_set_trace(None)
        tr = trace.cpu().numpy().reshape(nblk, 16)
        tr = tr[tr[:, 0] != 0]
        n = int((tr[0] != 0).sum())
        dt = np.diff(tr[:, :n], axis=1)
        names = ["input tile -> LDS + barrier A", "layer 1 (MFMA + h1 writes)", "barrier B", "layer 2 MFMAs (+ h2 half 0 writes)",
                 "barrier C + h2 half 1 + barrier D", "layer 3 MFMAs", "output transposition + stores", "barrier E"]
        print("trace (%d workgroups, second tile): cycles per phase, median / p90" % len(tr))
        for i in range(n - 1):
            print("  %-40s %7d %7d" % (names[i] if i < len(names) else "?", int(np.median(dt[:, i])), int(np.percentile(dt[:, i], 90))))
        print("  %-40s %7d" % ("one tile", int(np.median(tr[:, n - 1] - tr[:, 0]))))
