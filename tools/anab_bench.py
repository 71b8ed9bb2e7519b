#!/usr/bin/env python
"""m3d_anab_attend_bf16 alone at the plan's size (bs 64, 48x160 map, 337 keys): HIP-event time per launch, executed TFLOP/s.
M3D_ANAB_ONLINE=0: the two-pass form.      python tools/anab_bench.py [B]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from m3dssd_amd import _hip  # noqa: E402

L, dev = _hip.lib(), torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
HW, ck, ckp, cv, keys, kp = 48 * 160, 168, 192, 128, 337, 384
g = torch.Generator().manual_seed(1)
q = torch.zeros(B * HW, ckp)
q[:, :ck] = torch.randn(B * HW, ck, generator=g) * 0.5
khat = torch.zeros(B, kp, ckp)
khat[:, :keys, :ck] = torch.randn(B, keys, ck, generator=g) * 0.3
vhat = torch.zeros(B, cv, kp)
vhat[:, :, :keys] = torch.randn(B, cv, keys, generator=g)
res = torch.randn(B * HW, cv, generator=g)
dq, dk, dv, dr = (t.to(torch.bfloat16).contiguous().to(dev) for t in (q, khat, vhat, res))
sc, sh = torch.ones(cv, device=dev), torch.zeros(cv, device=dev)
out = torch.empty(B * HW, cv, device=dev, dtype=torch.bfloat16)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run():
    _hip.check(L.m3d_anab_attend_bf16(dq.data_ptr(), ckp, dk.data_ptr(), dv.data_ptr(), B, HW, ckp, keys, kp, cv, dr.data_ptr(), cv,
                                      sc.data_ptr(), sh.data_ptr(), 1, out.data_ptr(), cv, st))


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
online = os.environ.get("M3D_ANAB_ONLINE", "1") != "0"
gf = 2.0 * B * HW * 352 * (ckp * (1 if online else 2) + cv) / 1e9
print("bs %d: %.3f ms per launch, %.0f TFLOP/s executed (%s)" % (B, ms, gf / ms, "one pass" if online else "two passes"))
