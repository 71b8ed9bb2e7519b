"""GPU tests (-m gpu; every check goes through the C ABI of libm3dssd_hip.so) of the DCNv2 op (SURVEY 8 rows a2 / a3 and the drop-in m3d_dcn_v2_forward): the reference's known answer, the C oracle, the float64
numpy twin, closed forms, border grids, deformable groups, non-finite offsets, run-to-run identity.
Re-filed by component in round 5 (before: per-round files); tolerances are stated at the checks."""
import collections
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from m3dssd_amd import _hip, synth
from gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------ DCNv2
def test_dcn_zero_offset_identity_known_answer():
    """The reference's own op test, model/DCNv2/test.py:32-65, run through the drop-in modules."""
    from model.DCNv2.dcn_v2 import DCNv2
    dev = _dev()
    torch.manual_seed(0)
    N, C, H, W = 2, 2, 4, 4
    conv_offset = torch.nn.Conv2d(C, 18, 3, padding=1).to(dev)
    conv_mask = torch.nn.Conv2d(C, 9, 3, padding=1).to(dev)
    dcn = DCNv2(C, C, (3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).to(dev)
    with torch.no_grad():
        for m in (conv_offset, conv_mask):
            m.weight.zero_()
            m.bias.zero_()
        dcn.weight.zero_()
        dcn.bias.zero_()
        for c in range(C):
            dcn.weight[c, c, 1, 1] = 1.0
        x = torch.randn(N, C, H, W, device=dev)
        out = dcn(x, conv_offset(x), torch.sigmoid(conv_mask(x)))
    assert (x - out * 2).abs().max().item() < 1e-10


@pytest.mark.parametrize("shape", [(2, 2, 5, 6, 3, 3, 1, 1), (1, 128, 16, 40, 128, 3, 1, 1), (2, 256, 8, 20, 128, 3, 1, 1),
                                   (1, 512, 4, 10, 256, 3, 1, 1), (2, 128, 16, 40, 128, 1, 1, 0), (1, 20, 9, 7, 5, 3, 2, 1)])
def test_dcn_matches_oracle(shape):
    from m3dssd_amd.host import ops
    from oracle import dcn as odcn
    dev = _dev()
    n, c, h, w, co, k, stride, pad = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(co, c, k, k, generator=g) / (c * k * k) ** 0.5
    b = torch.randn(co, generator=g)
    ho, wo = odcn.out_size(h, w, k, k, stride, pad, 1)
    off = torch.randn(n, 2 * k * k, ho, wo, generator=g) * 3.0     # plenty of samples outside the map
    off[0, 0, 0, 0] = -1.0 + (1 if pad == 0 else 0) * 0.0            # exact -1 boundary (gate is strict)
    m = torch.rand(n, k * k, ho, wo, generator=g)
    ref = odcn.dcn_v2_forward(x, off, m, wt, b, stride, pad, 1, 1)
    got = ops.dcn_v2_forward(x.to(dev), off.to(dev), m.to(dev), wt.to(dev), b.to(dev), stride, pad, 1, 1).cpu()
    assert got.shape == ref.shape
    assert _relerr(got, ref) < 2e-4


def test_dcn_properties_closed_forms():
    """Closed forms of the modulated deformable convolution through the drop-in op (dcn_v2_im2col_cuda.cu:18-47,129-178):
    integer offsets == a shifted plain convolution of the zero-padded input; the output is linear in the mask; samples
    pushed entirely outside the map contribute exactly nothing (only the bias remains)."""
    from m3dssd_amd.host import ops
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    n, c, h, w, co, k, pad = 2, 32, 12, 14, 16, 3, 1
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(co, c, k, k, generator=g) / 17.0
    b = torch.randn(co, generator=g)
    ones = torch.ones(n, k * k, h, w)
    # integer offsets (dy, dx) = (2, -1) on every tap == conv of x shifted by (-2, +1) with zero fill
    off = torch.zeros(n, 2 * k * k, h, w)
    off[:, 0::2] = 2.0
    off[:, 1::2] = -1.0
    got = ops.dcn_v2_forward(x.to(dev), off.to(dev), ones.to(dev), wt.to(dev), b.to(dev), 1, pad, 1, 1).cpu()
    # tap (i, j) of output (ho, wo) reads x[ho - 1 + i + 2][wo - 1 + j - 1] (zero outside the map): crop one row at the top,
    # pad three at the bottom and two columns on the left, then a plain unpadded convolution
    ref = F.conv2d(F.pad(x, (2, 0, -1, 3)), wt, b)
    assert _relerr(got, ref) < 2e-4
    # linearity in the mask: f(a*m1 + b*m2) - bias == a*(f(m1) - bias) + b*(f(m2) - bias)
    offr = torch.randn(n, 2 * k * k, h, w, generator=g) * 2.0
    m1, m2 = torch.rand(n, k * k, h, w, generator=g), torch.rand(n, k * k, h, w, generator=g)
    f = lambda m: ops.dcn_v2_forward(x.to(dev), offr.to(dev), m.to(dev), wt.to(dev), b.to(dev), 1, pad, 1, 1).cpu() - b.view(1, -1, 1, 1)
    lhs, rhs = f(0.3 * m1 + 0.7 * m2), 0.3 * f(m1) + 0.7 * f(m2)
    assert (lhs - rhs).abs().max().item() < 1e-4 * max(1.0, rhs.abs().max().item())
    # every sample outside the map: only the bias is left, exactly
    far = torch.full((n, 2 * k * k, h, w), 100.0)
    out = ops.dcn_v2_forward(x.to(dev), far.to(dev), ones.to(dev), wt.to(dev), b.to(dev), 1, pad, 1, 1).cpu()
    assert torch.equal(out, b.view(1, -1, 1, 1).expand_as(out).contiguous())


def test_dcn_module_errors_like_reference():
    from model.DCNv2.dcn_v2 import DCNv2
    dev = _dev()
    dcn = DCNv2(4, 4, 3, 1, 1).to(dev)
    with pytest.raises(RuntimeError):       # channel mismatch (dcn_v2_cuda.c:37-39)
        dcn(torch.zeros(1, 3, 4, 4, device=dev), torch.zeros(1, 18, 4, 4, device=dev), torch.zeros(1, 9, 4, 4, device=dev))
    with pytest.raises(NotImplementedError):  # CPU input (dcn_v2_func.py:23-24)
        DCNv2(4, 4, 3, 1, 1)(torch.zeros(1, 4, 4, 4), torch.zeros(1, 18, 4, 4), torch.zeros(1, 9, 4, 4))


@pytest.mark.parametrize("shape", [(2, 32, 9, 13, 128, 3, 1, 1), (1, 64, 40, 52, 128, 3, 1, 1), (2, 128, 16, 24, 256, 1, 1, 0),
                                   (1, 48, 11, 7, 100, 3, 2, 1), (1, 32, 12, 20, 64, 3, 2, 1), (2, 64, 9, 9, 40, 1, 1, 0)])
@pytest.mark.parametrize("deform", [1, 0])
def test_dcn_wave_kernel_matches_block_kernel_and_oracle(shape, deform):
    """m3d_conv_wave_forward (register-resident, one wave per 32/64 px x 128 ch) vs the LDS-tiled igemm on the same
    descriptor, and vs the oracle im2col + GEMM: ragged M, borders, stride 2, channel padding, fused epilogue."""
    import ctypes
    from m3dssd_amd import _hip
    from m3dssd_amd.engine import pack_frag
    from m3dssd_amd.host import standalone as S
    from oracle import dcn as odcn
    dev = _dev()
    L = _hip.lib()
    n, ci, h, w, co, k, stride, pad = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, ci, h, w, generator=g)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    off = torch.randn(n, 2 * k * k, ho, wo, generator=g) * 1.5
    msk = torch.sigmoid(torch.randn(n, k * k, ho, wo, generator=g))
    if not deform:                                       # plain convolution = zero offsets, unit mask in the oracle
        off, msk = torch.zeros_like(off), torch.ones_like(msk)
    wt = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    b = torch.randn(co, generator=g)
    ref = odcn.dcn_v2_forward(x, off, msk, wt, b, stride, pad, 1, 1)
    res = torch.randn(n, co, ho, wo, generator=g)
    want = F.leaky_relu(ref + res, 0.01)
    cin_pad = (ci + 31) // 32 * 32
    cpt = 64 if co <= 64 else 128                        # Cout_pad 64 -> the 64-channel variant of the kernel
    v, _ = S._to_nhwc(x.to(dev), cin_pad)
    om, _ = S._to_nhwc(torch.cat([off, msk], 1).to(dev))
    rv, _ = S._to_nhwc(res.to(dev))
    out_blk, keep = S.conv_nhwc(v, wt.to(dev), b.to(dev), None, stride, pad, act=1, res=rv, om=om if deform else None,
                                cout_pad_to=cpt)
    blk = S._to_nchw(out_blk, co).cpu()
    wp, co_, cop, kh, kw = S._pack(wt.to(dev), cin_pad, cpt)
    frag = pack_frag(wp.view(cop, kh * kw * cin_pad), cop, dev)
    sc, sh = S._affine(co, b.to(dev), None, dev)
    out = torch.zeros(n * ho * wo * co, device=dev)
    d = _hip.ConvDesc()
    d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = v.ptr, v.cs, n, h, w, cin_pad
    d.wgt, d.Cout, d.Cout_pad = frag.data_ptr(), co, cop
    d.kh, d.kw, d.stride, d.pad, d.dil, d.Ho, d.Wo = k, k, stride, pad, 1, ho, wo
    d.out, d.out_cs, d.scale, d.shift = out.data_ptr(), co, sc.data_ptr(), sh.data_ptr()
    d.res, d.res_cs, d.res_mode, d.act, d.sigmoid_from = rv.ptr, rv.cs, 0, 1, -1
    if deform:
        d.dcn_offmask, d.dcn_om_cs = om.ptr, om.cs
    _hip.check(L.m3d_conv_wave_forward(ctypes.byref(d), S._stream()))
    got = out.view(n, ho, wo, co).permute(0, 3, 1, 2).cpu()
    assert _relerr(got, want) < 2e-4
    assert (got - blk).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    # split along K across waves (what the engine does for thin layers; with a workspace present m3d_conv_wave_forward splits
    # any layer that has fewer waves than the fill threshold): same result to fp32 reassociation, deterministic
    ws = torch.empty(8 * n * ho * wo * cop, device=dev)
    d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel() * 4
    outs = []
    for _ in range(2):
        out.zero_()
        _hip.check(L.m3d_conv_wave_forward(ctypes.byref(d), S._stream()))
        outs.append(out.clone())
    assert torch.equal(outs[0], outs[1])
    gs = outs[0].view(n, ho, wo, co).permute(0, 3, 1, 2).cpu()
    assert _relerr(gs, want) < 2e-4
    assert (gs - got).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    if k * k * cin_pad // 32 >= 8:                       # at least two splits of four steps: the short workspace is refused
        d.splitk_ws_bytes = 64
        assert L.m3d_conv_wave_forward(ctypes.byref(d), S._stream()) != 0 and b"workspace" in L.m3d_last_error()
    d.splitk_ws, d.splitk_ws_bytes = None, 0
    # sigmoid on channels >= 3 instead of the activation (the fused Q|K|V|S conv of ANAB): same epilogue as the block kernel
    d.sigmoid_from = 3
    out.zero_()
    _hip.check(L.m3d_conv_wave_forward(ctypes.byref(d), S._stream()))
    out_sg, _k = S.conv_nhwc(v, wt.to(dev), b.to(dev), None, stride, pad, act=1, res=rv, om=om if deform else None,
                             cout_pad_to=cpt, sigmoid_from=3)
    sg_blk = S._to_nchw(out_sg, co).cpu()
    sg_wave = out.view(n, ho, wo, co).permute(0, 3, 1, 2).cpu()
    assert (sg_wave - sg_blk).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    assert (sg_wave[:, 3:] >= 0).all() and (sg_wave[:, 3:] <= 1).all()


# ------------------------------------------------------------------------------------ DCNv2: independent pins
@pytest.mark.parametrize("case", [
    # n, c, h, w, co, k, stride, pad, dil
    (1, 8, 9, 11, 6, 3, 2, 1, 1), (2, 4, 10, 9, 5, 3, 1, 2, 2), (1, 12, 7, 8, 4, 3, 1, 0, 1), (2, 16, 6, 7, 8, 1, 1, 0, 1),
    (1, 5, 11, 13, 3, 3, 2, 0, 1), (1, 6, 12, 10, 7, 3, 2, 2, 2), (1, 3, 5, 5, 2, 1, 2, 0, 1),
])
def test_dcn_matches_numpy_float64_twin(case):
    """The HIP op against oracle.dcn.dcn_v2_forward_numpy -- per-output-pixel loops written from
    dcn_v2_im2col_cuda.cu:18-47,129-178 independently of the C restatement, float64 accumulation -- on the geometries the
    C-oracle tests do not reach (stride 2, dilation 2, pad 0, 1x1, odd sizes), offsets up to +-3 px and exact -1 gates."""
    from m3dssd_amd.host import ops
    from oracle import dcn as odcn
    dev = _dev()
    n, c, h, w, co, k, stride, pad, dil = case
    g = torch.Generator().manual_seed(sum(case) * 7 + 1)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(co, c, k, k, generator=g) / (c * k * k) ** 0.5
    b = torch.randn(co, generator=g)
    ho, wo = odcn.out_size(h, w, k, k, stride, pad, dil)
    off = torch.randn(n, 2 * k * k, ho, wo, generator=g) * 3.0
    off[0, 0, 0, 0] = -1.0 + pad                                   # tap 0 of pixel (0, 0): h_im exactly -1 (strict gate)
    off[0, 1, 0, 0] = float(w) + pad                               # w_im exactly W for the same tap: outside
    m = torch.rand(n, k * k, ho, wo, generator=g)
    ref = torch.from_numpy(odcn.dcn_v2_forward_numpy(x.numpy(), off.numpy(), m.numpy(), wt.numpy(), b.numpy(), stride, pad, dil))
    got = ops.dcn_v2_forward(x.to(dev), off.to(dev), m.to(dev), wt.to(dev), b.to(dev), stride, pad, dil, 1).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 2e-5 * (1.0 + ref.abs().max().item())
    # and the C restatement agrees with the numpy twin on the same geometry (the oracle's two forms pin each other)
    c_ref = odcn.dcn_v2_forward(x, off, m, wt, b, stride, pad, dil, 1)
    assert (c_ref - ref).abs().max().item() < 2e-5 * (1.0 + ref.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout", [(32, 128), (20, 24)])
def test_dcn_fp32_sampling_border_grid(cin, cout):
    """The corner in / out decisions of the DCNv2 sampling code (csrc/common.h dcn_corners) at exact border positions through
    the drop-in op: a 1x1 deformable conv whose pixel (i, j) samples at (hs[i % 8], ws[j % 8]) with hs / ws = -1 (out: the gate
    is strict), just inside, between rows, 0, the last row, past it (the high corners dropped), just below H, exactly H (out).
    (32, 128) runs the wave-granular kernel, (20, 24) the LDS-tiled implicit GEMM."""
    import torch

    from m3dssd_amd.host import ops
    from oracle import dcn as odcn
    dev = torch.device("cuda:0")
    h, w = 16, 24
    g = torch.Generator().manual_seed(cin)
    x = torch.randn(1, cin, h, w, generator=g) + 3.0
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    b = torch.zeros(cout)
    hs = [-1.0, -0.999, -0.5, 0.0, h - 1.0, h - 0.5, h - 0.001, float(h)]
    ws = [-1.0, -0.999, -0.5, 0.0, w - 1.0, w - 0.5, w - 0.001, float(w)]
    off = torch.zeros(1, 2, h, w)
    for i in range(h):
        for j in range(w):
            off[0, 0, i, j] = hs[i % 8] - i
            off[0, 1, i, j] = ws[j % 8] - j
    m = torch.ones(1, 1, h, w)
    ref = odcn.dcn_v2_forward(x, off, m, wt, b, 1, 0, 1, 1)
    got = ops.dcn_v2_forward(x.to(dev), off.to(dev), m.to(dev), wt.to(dev), b.to(dev), 1, 0, 1, 1).cpu()
    assert (got - ref).abs().max().item() < 2e-4 * ref.abs().max().item()
    for sl in (got[0, :, 0::8, :], got[0, :, 7::8, :], got[0, :, :, 0::8], got[0, :, :, 7::8]):
        assert torch.equal(sl, torch.zeros_like(sl))                 # sampled exactly at -1 / H / W: nothing at all


@pytest.mark.gpu
def test_dcn_wave_fp32_run_to_run_identical_at_large_grids():
    """Run-to-run identity of the fp32 deformable wave kernel at a full-size grid (15360 waves, two per SIMD): the bf16 kernel's
    sampling code, written with compares, dropped a corner in lanes 48-63 of a wave once per 10^5..10^6 states (DESIGN.md
    section 3); all three deformable kernels now build the state through `dcn_corners` (csrc/common.h, no SGPR lane masks).
    12 launches on the same operands, bit-identical outputs."""
    import ctypes

    import torch

    from m3dssd_amd import _hip
    from m3dssd_amd.engine import pack_frag
    dev = torch.device("cuda:0")
    L = _hip.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for cin, cout, h, w, b, k in [(128, 128, 48, 160, 64, 3), (128, 128, 48, 160, 32, 1)]:
        g = torch.Generator().manual_seed(cin + k)
        x = torch.randn(b * h * w * cin, generator=g).to(dev)
        wf = pack_frag(torch.randn(cout, k * k * cin, generator=g) / (k * k * cin) ** 0.5, cout, dev)
        kk = k * k
        om = torch.cat([torch.randn(b * h * w, 2 * kk, generator=g) * 2.0, torch.rand(b * h * w, kk, generator=g),
                        torch.zeros(b * h * w, 28 - 3 * kk)], 1).contiguous().to(dev)
        outs = []
        for _ in range(12):
            out = torch.zeros(b * h * w * cout, device=dev)
            d = _hip.ConvDesc()
            d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.data_ptr(), cin, b, h, w, cin
            d.wgt, d.Cout, d.Cout_pad = wf.data_ptr(), cout, cout
            d.kh = d.kw = k
            d.stride, d.pad, d.dil, d.Ho, d.Wo = 1, k // 2, 1, h, w
            d.out, d.out_cs, d.act, d.sigmoid_from = out.data_ptr(), cout, 1, -1
            d.dcn_offmask, d.dcn_om_cs = om.data_ptr(), 28
            _hip.check(L.m3d_conv_wave_forward(ctypes.byref(d), st))
            torch.cuda.synchronize()
            outs.append(out)
        assert torch.isfinite(outs[0]).all()
        for o in outs[1:]:
            assert torch.equal(outs[0], o), (cin, k, int((outs[0] != o).sum()))


# ------------------------------------------------------------------------------------ DCN module contract
@pytest.mark.parametrize("shape", [(2, 64, 32, 40, 64, 3, 1, 1, 2), (1, 48, 9, 11, 20, 3, 2, 1, 4), (2, 64, 16, 16, 32, 1, 1, 0, 2)])
def test_dcn_v2_deformable_groups_match_oracle(shape):
    """deformable_groups > 1 through the drop-in op: group g's channels sample at group g's offsets / masks
    (dcn_v2_im2col_cuda.cu:139-156)."""
    from m3dssd_amd.host import ops
    from oracle import dcn as odcn
    dev = _dev()
    n, c, h, w, co, k, stride, pad, G = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(co, c, k, k, generator=g) / (c * k * k) ** 0.5
    b = torch.randn(co, generator=g)
    ho, wo = odcn.out_size(h, w, k, k, stride, pad, 1)
    off = torch.randn(n, G * 2 * k * k, ho, wo, generator=g) * 2.0
    m = torch.rand(n, G * k * k, ho, wo, generator=g)
    ref = odcn.dcn_v2_forward(x, off, m, wt, b, stride, pad, 1, G)
    got = ops.dcn_v2_forward(x.to(dev), off.to(dev), m.to(dev), wt.to(dev), b.to(dev), stride, pad, 1, G).cpu()
    assert got.shape == ref.shape and _relerr_t(got, ref) < 2e-4
    # the groups really differ: feeding group 0's offsets to every group changes the result
    off0 = off[:, :2 * k * k].repeat(1, G, 1, 1)
    assert _relerr_t(odcn.dcn_v2_forward(x, off0, m, wt, b, stride, pad, 1, G), ref) > 1e-2
    with pytest.raises(RuntimeError):
        ops.dcn_v2_forward(x.to(dev), off.to(dev), m.to(dev), wt.to(dev), b.to(dev), stride, pad, 1, 5)


def test_dcn_module_example_of_the_reference_with_two_groups():
    """model/DCNv2/test.py:169-179 `example_dconv`: DCN(64, 64, (3, 3), 1, 1, deformable_groups=2) on 2 x 64 x 128 x 128."""
    from model.DCNv2.dcn_v2 import DCN
    from oracle import dcn as odcn
    dev = _dev()
    torch.manual_seed(3)
    dcn = DCN(64, 64, kernel_size=(3, 3), stride=1, padding=1, deformable_groups=2)
    assert tuple(dcn.conv_offset_mask.weight.shape) == (54, 64, 3, 3)
    with torch.no_grad():
        dcn.conv_offset_mask.weight.normal_(0, 0.02)
        dcn.conv_offset_mask.bias.normal_(0, 0.5)
        dcn.bias.normal_(0, 0.1)
    x = torch.randn(2, 64, 128, 128)
    with torch.no_grad():
        out = torch.nn.functional.conv2d(x, dcn.conv_offset_mask.weight, dcn.conv_offset_mask.bias, padding=1)
        o1, o2, mask = torch.chunk(out, 3, dim=1)                              # dcn_v2.py:65-68
        ref = odcn.dcn_v2_forward(x, torch.cat((o1, o2), 1), torch.sigmoid(mask), dcn.weight, dcn.bias, 1, 1, 1, 2)
        got = dcn.to(dev)(x.to(dev)).cpu()
    assert tuple(got.shape) == (2, 64, 128, 128)
    assert _relerr_t(got, ref) < 5e-4


def test_dcn_non_finite_sampling_positions_contribute_nothing():
    """`h_im > -1 && w_im > -1 && h_im < H && w_im < W` (dcn_v2_im2col_cuda.cu:165) is false for a NaN coordinate: the tap is
    skipped.  csrc/common.h dcn_corners decides with fminf / fmaxf (which drop NaNs) and therefore carries an explicit
    non-finite term; +-inf positions are outside by the ordinary test."""
    from m3dssd_amd.host import ops
    from oracle import dcn as odcn
    dev = _dev()
    g = torch.Generator().manual_seed(77)
    n, c, h, w, co, k = 1, 32, 8, 16, 32, 3
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(co, c, k, k, generator=g) / 17.0
    b = torch.randn(co, generator=g)
    off = torch.randn(n, 2 * k * k, h, w, generator=g)
    m = torch.rand(n, k * k, h, w, generator=g)
    nan, inf = float("nan"), float("inf")
    off[0, 0, 2, 3] = nan          # tap 0: dh NaN, dw finite
    off[0, 3, 2, 4] = nan          # tap 1: dw NaN, dh finite
    off[0, 8, 5, 5] = nan
    off[0, 9, 5, 5] = nan          # tap 4: both
    off[0, 10, 6, 6] = inf
    off[0, 13, 6, 7] = -inf
    ref = odcn.dcn_v2_forward(x, off, m, wt, b, 1, 1, 1, 1)
    assert torch.isfinite(ref).all()
    got = ops.dcn_v2_forward(x.to(dev), off.to(dev), m.to(dev), wt.to(dev), b.to(dev), 1, 1, 1, 1).cpu()
    assert torch.isfinite(got).all()
    assert _relerr_t(got, ref) < 2e-4

