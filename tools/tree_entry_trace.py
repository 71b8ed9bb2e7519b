"""bf16 tree-entry kernel (max-pool + project + stride-2 conv1): launch time per DLA level, and -- with the diagnostic build
`make -C m3dssd_amd/csrc trace` -- the timeline of the SECOND (item, chunk) of every persistent workgroup (thread 0: top of the iteration | tile staged (barrier) |
MFMAs done | stores issued).      python tools/tree_entry_trace.py [B]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from m3dssd_amd import _hip                               # noqa: E402
from m3dssd_amd.engine_bf16 import pack_tree_entry        # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
L = _hip.lib()
tp = "m3dssd_amd/csrc/build/libm3dssd_hip_trace.so"
T = ctypes.CDLL(tp) if os.path.exists(tp) else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, cin, H, W, with_bottom in (("level2", 32, 192, 640, False), ("level3", 64, 96, 320, True), ("level4", 128, 48, 160, True),
                                     ("level5", 256, 24, 80, True)):
    co = 2 * cin
    g = torch.Generator().manual_seed(cin)
    x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).to(dev)
    wf = pack_tree_entry(torch.randn(co, cin, 3, 3, generator=g) / (9 * cin) ** 0.5, torch.ones(co), torch.randn(co, cin, 1, 1, generator=g) / cin ** 0.5,
                         torch.ones(co), dev)
    z = torch.zeros(co, device=dev)
    t = torch.empty(B, H // 2, W // 2, co, device=dev, dtype=torch.bfloat16)
    r = torch.empty_like(t)
    bt = torch.empty(B, H // 2, W // 2, cin, device=dev, dtype=torch.bfloat16)
    d = _hip.TreeEntryBf16Desc()
    d.inp, d.in_cs, d.N, d.H, d.W, d.Cin, d.Cout = x.data_ptr(), cin, B, H, W, cin, co
    d.wfrag, d.shift1, d.shiftp = wf.data_ptr(), z.data_ptr(), z.data_ptr()
    d.t, d.t_cs, d.res, d.res_cs = t.data_ptr(), co, r.data_ptr(), co
    if with_bottom:
        d.bottom, d.bottom_cs = bt.data_ptr(), cin
    for _ in range(3):
        _hip.check(L.m3d_tree_entry_bf16_forward(ctypes.byref(d), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _hip.check(L.m3d_tree_entry_bf16_forward(ctypes.byref(d), st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    nbytes = B * (H * W * cin + (H // 2) * (W // 2) * (2 * co + (cin if with_bottom else 0))) * 2
    print("%s (Cin %d, %dx%d): %.4f ms, %.2f TB/s of algorithmic bytes, %.0f TFLOP/s" %
          (name, cin, H, W, ms, nbytes / ms / 1e9, 2.0 * B * (H // 2) * (W // 2) * co * cin * 10 / ms / 1e9))
    if T is not None and hasattr(T, "m3d_tree_entry_set_trace"):
        T.m3d_tree_entry_bf16_forward.argtypes = L.m3d_tree_entry_bf16_forward.argtypes
        T.m3d_tree_entry_set_trace.argtypes = [ctypes.c_void_p]
        nblk = 512
        trace = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
        assert T.m3d_tree_entry_bf16_forward(ctypes.byref(d), st) == 0
        torch.cuda.synchronize()
        T.m3d_tree_entry_set_trace(trace.data_ptr())
        assert T.m3d_tree_entry_bf16_forward(ctypes.byref(d), st) == 0
        torch.cuda.synchronize()
        T.m3d_tree_entry_set_trace(None)
        tr = trace.cpu().numpy().reshape(nblk, 8)
        tr = tr[tr[:, 0] != 0]
        n = int((tr[0] != 0).sum())
        dt = np.diff(tr[:, :n], axis=1)
        print("   cycles (median over %d workgroups): " % len(tr) + "  ".join(
            "%s %d" % (nm, int(np.median(dt[:, i]))) for i, nm in enumerate(["weights requested + convert + LDS + barrier", "next tile requested + MFMAs", "stores"][:n - 1]))
              + "  | life %d" % int(np.median(tr[:, n - 1] - tr[:, 0])))
