"""Fused bf16 front end (stem -> level0 -> level1): launch time, and -- with the diagnostic build
`make -C m3dssd_amd/csrc trace` -- the in-kernel timeline of bf16_frontend2_kernel (thread 0 of every workgroup: start | image
patch staged | stem done | level0 done | level1 stored; the SECOND tile of every persistent workgroup).
    python tools/front2_trace.py [B]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from m3dssd_amd import _hip                               # noqa: E402
from m3dssd_amd.engine_bf16 import pack_frontend_f16   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, W = 384, 1280
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
ws, w0, w1 = (torch.randn(16, 3, 7, 7, generator=g) / 12, torch.randn(16, 16, 3, 3, generator=g) / 12, torch.randn(32, 16, 3, 3, generator=g) / 12)
aff = [(torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1) for c in (16, 16, 32)]
img = torch.randn(B, 3, H, W, generator=g).to(dev)
out = torch.empty(B, H // 2, W // 2, 32, device=dev, dtype=torch.bfloat16)
f2 = pack_frontend_f16(ws, aff[0], w0, aff[1], w1, aff[2], dev)
mean3, stds3 = (ctypes.c_float * 3)(0.5, 0.5, 0.5), (ctypes.c_float * 3)(0.2, 0.2, 0.2)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def form2(L):
    assert L.m3d_frontend2_bf16_forward(img.data_ptr(), 0, 0, 0, mean3, stds3, f2[0].data_ptr(), f2[1].data_ptr(), f2[2].data_ptr(),
                                        f2[3].data_ptr(), f2[4].data_ptr(), f2[5].data_ptr(), out.data_ptr(), 32, B, H, W, st) == 0


def timeit(fn, L, reps=20):
    for _ in range(3):
        fn(L)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn(L)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


L = _hip.lib()
print("B = %d, back to back (persistent, 2 workgroups per CU): %.4f ms" % (B, timeit(form2, L)))
tp = "m3dssd_amd/csrc/build/libm3dssd_hip_trace.so"
if os.path.exists(tp):
    T = ctypes.CDLL(tp)
    if hasattr(T, "m3d_front2_set_trace"):
        T.m3d_frontend2_bf16_forward.argtypes = L.m3d_frontend2_bf16_forward.argtypes
        T.m3d_front2_set_trace.argtypes = [ctypes.c_void_p]
        nblk = 256 * 2
        for occ in ("2",):
            trace = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
            form2(T)
            torch.cuda.synchronize()
            T.m3d_front2_set_trace(trace.data_ptr())
            form2(T)
            torch.cuda.synchronize()
            T.m3d_front2_set_trace(None)
            t = trace.cpu().numpy().reshape(nblk, 8)
            t = t[t[:, 0] != 0]
            dt = np.diff(t[:, :5], axis=1)
            print("trace, %s workgroups per CU (%d workgroups): cycles per phase, median / p90" % (occ, len(t)))
            for i, nm in enumerate(["image patch -> LDS", "stem", "level0", "level1 + stores"]):
                print("  %-22s %7d %7d" % (nm, int(np.median(dt[:, i])), int(np.percentile(dt[:, i], 90))))
            print("  %-22s %7d" % ("workgroup life", int(np.median(t[:, 4] - t[:, 0]))))
