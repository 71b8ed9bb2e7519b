#!/usr/bin/env python
"""ANAB pyramid pooling alone at the plan's size (K|V map of 48x160 pixels x 296 channels, bf16 at bs 64 / fp32 at bs 8):
HIP-event time of the two launches (pool + finish) and the rate over the bytes the first one has to read once.
    python tools/anab_pool_bench.py [B] [f32]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from m3dssd_amd import _hip  # noqa: E402

L, dev = _hip.lib(), torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
f32 = len(sys.argv) > 2 and sys.argv[2] == "f32"
H, W, ck, cv, kp, ckp = 48, 160, 168, 128, 352, 192
C = ck + cv
g = torch.Generator().manual_seed(2)
kv = torch.randn(B * H * W, C + 8, generator=g).to(torch.float32 if f32 else torch.bfloat16).to(dev)
s = torch.rand(B * H * W, 8, generator=g).to(dev)
scratch = torch.empty(L.m3d_anab_pool_nested_scratch_bytes(B, C) // 4, device=dev)
khat = torch.zeros(B * kp * ckp, device=dev)
vhat = torch.zeros(B * cv * kp, device=dev)
k16, v16 = torch.zeros_like(khat, dtype=torch.bfloat16), torch.zeros_like(vhat, dtype=torch.bfloat16)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run():
    if f32:
        _hip.check(L.m3d_anab_pool_nested(kv.data_ptr(), C + 8, s.data_ptr(), 8, B, H, W, ck, cv, scratch.data_ptr(), khat.data_ptr(), kp, ckp,
                                          vhat.data_ptr(), 0, st))
    else:
        _hip.check(L.m3d_anab_pool_nested_bf16_ex(kv.data_ptr(), C + 8, s.data_ptr(), 8, B, H, W, ck, cv, scratch.data_ptr(), khat.data_ptr(),
                                                  kp, ckp, vhat.data_ptr(), 0, k16.data_ptr(), v16.data_ptr(), st))


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
nbytes = B * H * W * (C * (4 if f32 else 2) + 16)
print("bs %d %s: %.4f ms for pool + finish; %.0f MB of features + gates -> %.2f TB/s over both launches; checksum %.6f"
      % (B, "f32" if f32 else "bf16", ms, nbytes / 1e6, nbytes / ms / 1e9,
         float(khat.double().sum() + vhat.double().sum())))
