#!/usr/bin/env python
"""Would a third wave per SIMD pay for the fp32 deformable kernel?

`conv_wave_kernel<true, 4>` (csrc/dcn_wave.hip) needs 216 registers (246 before the K order became a compile-time constant): two waves per SIMD.  64 of them hold the 16 gathered corner
chunks of the step in flight.  The review's proposal: move those to LDS so that three waves fit (<= 168 registers).  LDS cannot
take them (16 KB per wave and step, 12 waves per CU), so before any redesign this experiment asks the question the proposal rests
on: WITH the registers freed, is the kernel faster at three waves per SIMD than at two?

The diagnostic library (`make -C m3dssd_amd/csrc trace`) carries variants of the kernel that gather 1 or 2 of the 4 corners per
(pixel, tap) -- WRONG results, same prologue / K loop / MFMA stream / epilogue, 48 / 32 fewer registers -- each built with a
register bound of two and of three waves per SIMD (`m3d_conv_wave_set_variant`):
    0  as built (4 corners, 2 waves, 216 VGPRs)        5  as built, bound to 3 waves (168 VGPRs, 177 spilled)
    1  1 corner, 2 waves (172)                          2  1 corner, 3 waves (168, no spill)
    3  2 corners, 2 waves (186)                         4  2 corners, 3 waves (168, 18 spilled)
Reading: (1 - 2) is what the third wave buys when registers are free; (0 - 1) is what the gather itself costs at two waves.

    python tools/conv_wave_third_wave.py [reps]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, ".")
from m3dssd_amd import _hip                       # noqa: E402
from m3dssd_amd.engine import pack_frag           # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tp = "m3dssd_amd/csrc/build/libm3dssd_hip_trace.so"
if not os.path.exists(tp):
    raise SystemExit("build the diagnostic library first: make -C m3dssd_amd/csrc trace")
T = ctypes.CDLL(tp)
T.m3d_conv_wave_forward.argtypes = [ctypes.POINTER(_hip.ConvDesc), ctypes.c_void_p]
T.m3d_conv_wave_set_variant.argtypes = [ctypes.c_int]
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
NAMES = {0: "as built: 4 corners, 2 waves/SIMD (216 VGPR)", 1: "1 corner, 2 waves (172)", 2: "1 corner, 3 waves (168)",
         3: "2 corners, 2 waves (186)", 4: "2 corners, 3 waves (168, 18 spilled)", 5: "4 corners, 3 waves (168, 177 spilled)"}


def layer(B, H, W, cin, cout, sigma):
    g = torch.Generator(device="cpu").manual_seed(H * 7 + cin)
    x = torch.randn(B * H * W * cin, generator=g).to(dev)
    out = torch.empty(B * H * W * cout, device=dev)
    om = torch.cat([torch.randn(B * H * W, 18, generator=g) * sigma, torch.rand(B * H * W, 9, generator=g),
                    torch.zeros(B * H * W, 1)], 1).contiguous().to(dev)
    wf = pack_frag(torch.randn(cout, 9 * cin, generator=g) / (9 * cin) ** 0.5, cout, dev)
    d = _hip.ConvDesc()
    d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.data_ptr(), cin, B, H, W, cin
    d.Cout, d.Cout_pad, d.wgt = cout, cout, wf.data_ptr()
    d.kh = d.kw = 3
    d.stride, d.pad, d.dil, d.Ho, d.Wo = 1, 1, 1, H, W
    d.out, d.out_cs, d.act, d.sigmoid_from = out.data_ptr(), cout, 1, -1
    d.dcn_offmask, d.dcn_om_cs = om.data_ptr(), 28
    return d, (x, out, om, wf)


def timeit(d, variant):
    T.m3d_conv_wave_set_variant(variant)
    for _ in range(5):
        assert T.m3d_conv_wave_forward(ctypes.byref(d), st) == 0
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            T.m3d_conv_wave_forward(ctypes.byref(d), st)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    T.m3d_conv_wave_set_variant(0)
    return best


# the deformable layers of the fp32 plan at bs 8 that run unsplit (128 -> 128 @ 48x160) and the 24x80 node (256 -> 256)
for B, H, W, cin, cout, sigma in [(8, 48, 160, 128, 128, 1.0), (8, 48, 160, 128, 128, 3.0), (8, 24, 80, 256, 256, 1.0)]:
    d, keep = layer(B, H, W, cin, cout, sigma)
    fl = 2.0 * B * H * W * cout * 9 * cin
    print("layer %d x %dx%d, %d -> %d, offsets ~ N(0, %.0f^2): %.2f GFLOP" % (B, H, W, cin, cout, sigma, fl / 1e9))
    res = {}
    for v in (0, 1, 2, 3, 4, 5):
        ms = timeit(d, v)
        res[v] = ms
        print("  variant %d  %-44s %.4f ms  %6.1f TFLOP/s  (%.3f of 157.3)" % (v, NAMES[v], ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3))
    print("  third wave with the registers freed (1 corner): %+.1f %%;  (2 corners, 18 spills): %+.1f %%;  gather cost at 2 waves: "
          "4 -> 1 corners %+.1f %%" % (100 * (res[2] / res[1] - 1), 100 * (res[4] / res[3] - 1), 100 * (res[1] / res[0] - 1)))
