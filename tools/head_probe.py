"""Time the fused head-MLP kernel alone (HIP events): python tools/head_probe.py [heads] [B]
Env M3D_ABLATE_MLP selects the diagnostic ablations documented in csrc/head_mlp.hip."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, ".")
from m3dssd_amd import _hip                       # noqa: E402
from m3dssd_amd.engine import pack_frag           # noqa: E402

heads = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
HW, cin, cout, cpad = 48 * 160, 128, 36, 64
M = B * HW
L = _hip.lib()
TR = os.environ.get("HEAD_TRACE")
if TR:
    L = ctypes.CDLL("m3dssd_amd/csrc/build/libm3dssd_hip_trace.so")
    L.m3d_head_mlp_forward_batched.argtypes = [ctypes.POINTER(_hip.MlpDesc), ctypes.c_int, ctypes.c_void_p]
    L.m3d_head_set_trace.argtypes = [ctypes.c_void_p]
keep = []
arr = (_hip.MlpDesc * heads)()
x = torch.randn(M, cin, device=dev)
for i in range(heads):
    d = arr[i]
    d.inp, d.in_cs, d.M, d.Cin = x.data_ptr(), cin, M, cin
    for slot, (co, ci, pad) in zip("123", [(256, cin, 256), (256, 256, 256), (cout, 256, cpad)]):
        w = pack_frag(torch.randn(co, ci) / ci ** 0.5, pad, dev)
        sc, sh = torch.ones(pad, device=dev), torch.zeros(pad, device=dev)
        keep += [w, sc, sh]
        setattr(d, "w" + slot, w.data_ptr()); setattr(d, "s" + slot, sc.data_ptr()); setattr(d, "t" + slot, sh.data_ptr())
    out = torch.zeros(B, cout, HW, device=dev)
    keep.append(out)
    d.Cout, d.Cout_pad, d.out, d.out_img_stride, d.HW = cout, cpad, out.data_ptr(), cout * HW, HW
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    _hip.check(L.m3d_head_mlp_forward_batched(arr, heads, st))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    _hip.check(L.m3d_head_mlp_forward_batched(arr, heads, st))
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
fl = 2.0 * M * (cin * 256 + 256 * 256 + 256 * cpad) * heads
print("heads=%d B=%d  %.4f ms  executed %.1f TFLOP/s (%.1f%% of 157.3)" % (heads, B, ms, fl / ms / 1e9, fl / ms / 1e9 / 1.573))

if TR:
    import numpy as np
    nblk = (M + 63) // 64 * heads
    trace = torch.zeros(nblk * 4 * 64, dtype=torch.int64, device=dev)
    L.m3d_head_set_trace(trace.data_ptr())
    _hip.check(L.m3d_head_mlp_forward_batched(arr, heads, st))
    torch.cuda.synchronize()
    t = trace.cpu().numpy().reshape(nblk, 4, 64)
    dur = t.max(axis=(1, 2)) - t[:, 0, 0]
    print("block durations: min %d median %d max %d ticks" % (dur.min(), int(np.median(dur)), dur.max()))
    for blk in (nblk // 2, nblk // 2 + 1):
        for wv in (0,):
            s_ = t[blk, wv]
            s_ = s_[s_ > 0] - t[blk, 0, 0]
            print("block %d wave %d:" % (blk, wv), " ".join("%d" % v for v in s_))
            print("   deltas:", " ".join("%d" % v for v in np.diff(s_)))
