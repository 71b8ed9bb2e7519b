"""In-kernel timeline of bf16_conv_kernel (diagnostic build `make -C m3dssd_amd/csrc trace`, -DBF16_TRACE):
    python tools/bf16_conv_trace.py [Cin] [Cout] [H] [W] [B] [k] [offset std: deformable mode] [v: per-step table]
Per K-step of wave 0 of every workgroup: cycles from loop top to loads issued, to MFMAs issued, to staging writes done
(vmcnt waits + ds_write), to barrier passed."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from m3dssd_amd import _hip                               # noqa: E402
from m3dssd_amd.engine_bf16 import pack_conv_bf16          # noqa: E402

cin = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 256
H = int(sys.argv[3]) if len(sys.argv) > 3 else 24
W = int(sys.argv[4]) if len(sys.argv) > 4 else 80
B = int(sys.argv[5]) if len(sys.argv) > 5 else 64
k = int(sys.argv[6]) if len(sys.argv) > 6 else 3
deform = float(sys.argv[7]) if len(sys.argv) > 7 else -1.0      # >= 0: deformable mode, offsets ~ N(0, deform)
dev = torch.device("cuda:0")
L = ctypes.CDLL("m3dssd_amd/csrc/build/libm3dssd_hip_trace.so")
L.m3d_conv_bf16_forward.argtypes = [ctypes.POINTER(_hip.ConvBf16Desc), ctypes.c_void_p]
L.m3d_bf16_conv_set_trace.argtypes = [ctypes.c_void_p]
x = torch.randn(B * H * W * cin, device=dev).to(torch.bfloat16)
wp, kpad = pack_conv_bf16(torch.randn(cout, cin, k, k) / (k * k * cin) ** 0.5, None, None, dev)
out = torch.empty(B * H * W * cout, device=dev, dtype=torch.bfloat16)
d = _hip.ConvBf16Desc()
d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.data_ptr(), cin, B, H, W, cin
d.wgt, d.Cout, d.Cout_pad, d.Kpad = wp.data_ptr(), cout, wp.shape[0], kpad
d.kh = d.kw = k
d.stride, d.pad, d.Ho, d.Wo = 1, k // 2, H, W
d.out, d.out_cs, d.out_mode, d.act, d.sigmoid_from, d.groups = out.data_ptr(), cout, 0, 1, -1, 1
if deform >= 0:
    om = torch.cat([torch.randn(B * H * W, 2 * k * k, device=dev) * deform, torch.rand(B * H * W, k * k, device=dev),
                    torch.zeros(B * H * W, 32 - 3 * k * k, device=dev)], 1).contiguous()
    d.dcn_offmask, d.dcn_om_cs = om.data_ptr(), 32
bn = 128 if wp.shape[0] % 128 == 0 else (64 if wp.shape[0] % 64 == 0 else 32)
grid = (B * H * W + 127) // 128 * (wp.shape[0] // bn)          # upper bound: the halo-tile kernels launch fewer, larger workgroups
L.m3d_conv_bf16_variant.argtypes = [ctypes.POINTER(_hip.ConvBf16Desc)]
trace = torch.zeros(grid * 160, dtype=torch.int64, device=dev)
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
ms = e0.elapsed_time(e1)
KT = kpad // 64
t = trace.cpu().numpy().reshape(grid, 160)
variant = L.m3d_conv_bf16_variant(ctypes.byref(d))
if variant:
    grid = ((H + 7) // 8) * ((W + 16 * variant - 1) // (16 * variant)) * B * (wp.shape[0] // bn)
    t = t[:grid]
    print("halo-tile kernel, 8 x %d patches" % (16 * variant))
fl = 2.0 * B * H * W * cout * k * k * cin
n = min(KT, 38)
dur = t[:, 1 + 4 * n] - t[:, 0] if 1 + 4 * n < 160 else t[:, 1 + 4 * (n - 1)] - t[:, 0]
print("grid %d workgroups, KT %d, launch %.4f ms (%.1f TFLOP/s); workgroup (first %d K-steps) ticks: min %d median %d max %d"
      % (grid, KT, ms, fl / ms / 1e9, n, dur.min(), int(np.median(dur)), dur.max()))
if (t[:, 159] != 0).all():
    cyc, real = t[:, 158] - t[:, 156], t[:, 159] - t[:, 157]
    ok = real > 0
    print("shader clock seen by the workgroups (s_memtime / s_memrealtime at 100 MHz): median %.2f GHz (p10 %.2f, p90 %.2f)"
          % tuple(np.percentile(cyc[ok] / real[ok] * 0.1, [50, 10, 90])))
    span = (t[:, 159].max() - t[:, 157].min()) / 100.0
    print("first workgroup start -> last K loop end: %.1f us of the %.1f us launch" % (span, ms * 1000))
s = t[:, 1:1 + 4 * n].reshape(grid, n, 4)
prolog = t[:, 1] - t[:, 0]
top = s[:, :, 0]
issue = s[:, :, 1] - s[:, :, 0]
mfma = s[:, :, 2] - s[:, :, 1]
store = s[:, :, 3] - s[:, :, 2]
nxt = np.concatenate([s[:, 1:, 0], t[:, 1 + 4 * n:2 + 4 * n]], 1) if 1 + 4 * n < 160 else None
print("prologue (first tile loaded + staged): median %d ticks" % int(np.median(prolog)))
print("per K-step medians (ticks of the 100 MHz? s_memtime counter = shader cycles): load issue %d | MFMA section %d | vmcnt wait + ds_write %d%s"
      % (int(np.median(issue)), int(np.median(mfma)), int(np.median(store)),
         (" | barrier %d" % int(np.median(nxt - s[:, :, 3]))) if nxt is not None else ""))
print("K-step period median %d" % int(np.median(np.diff(top, axis=1))))
if 5 + 4 * n < 160 and (t[:, 5 + 4 * n] != 0).all():
    b = 1 + 4 * n
    print("epilogue phases: affine -> LDS + barrier %d | accumulators -> LDS tile %d | barrier %d | tile -> global stores issued %d"
          % tuple(int(np.median(t[:, b + k + 1] - t[:, b + k])) for k in range(4)))
elif 2 + 4 * n < 160 and (t[:, 2 + 4 * n] != 0).all():
    print("epilogue (accumulators -> affine / activation -> stores issued): median %d ticks; workgroup start -> end median %d"
          % (int(np.median(t[:, 2 + 4 * n] - t[:, 1 + 4 * n])), int(np.median(t[:, 2 + 4 * n] - t[:, 0]))))
if len(sys.argv) > 8:       # per-K-step medians
    for kk in range(n):
        print("step %2d: issue %5d mfma %5d store %5d%s" % (kk, int(np.median(issue[:, kk])), int(np.median(mfma[:, kk])),
              int(np.median(store[:, kk])), (" barrier %5d" % int(np.median((nxt - s[:, :, 3])[:, kk]))) if nxt is not None else ""))
