"""GPU tests (-m gpu; every check goes through the C ABI of libm3dssd_hip.so) of the fp32 convolution kernels and fused blocks: implicit GEMM, split-K, Winograd F(2x2) / F(4x4),
the level0 kernel, fused head MLPs, ANAB pooling -- each against torch fp32 / fp64.
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


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_igemm_matches_torch(case):
    from m3dssd_amd.host import standalone as S
    dev = _dev()
    n, ci, h, w, co, k, stride, pad, bias, bn, act, res = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    b = torch.randn(co, generator=g) if bias else None
    bnm = None
    ref = F.conv2d(x, wt, b, stride=stride, padding=pad)
    if bn:
        bnm = torch.nn.BatchNorm2d(co).eval()
        with torch.no_grad():
            bnm.weight.uniform_(0.5, 1.5, generator=g)
            bnm.bias.normal_(0, 0.2, generator=g)
            bnm.running_mean.normal_(0, 0.2, generator=g)
            bnm.running_var.uniform_(0.5, 1.5, generator=g)
        ref = bnm(ref)
    r = None
    if res:
        r = torch.randn_like(ref)
        ref = ref + r
    if act:
        ref = F.leaky_relu(ref, 0.01)
    with torch.no_grad():
        v, _ = S._to_nhwc(x.to(dev))
        rv = S._to_nhwc(r.to(dev))[0] if res else None
        out, keep = S.conv_nhwc(v, wt.to(dev), None if b is None else b.to(dev), None if bnm is None else bnm.to(dev),
                                stride, pad, act=act, res=rv)
        got = S._to_nchw(out, co).cpu()
    assert got.shape == ref.shape
    assert _relerr(got, ref.detach()) < 2e-4
    with torch.no_grad():                      # small-M cases split along K by default: the unsplit launch must agree
        out1, keep1 = S.conv_nhwc(v, wt.to(dev), None if b is None else b.to(dev), None if bnm is None else bnm.to(dev),
                                  stride, pad, act=act, res=rv, splitk=False)
        one = S._to_nchw(out1, co).cpu()
    assert _relerr(one, ref.detach()) < 2e-4
    assert (one - got).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


def test_conv_splitk_plan_and_workspace_check():
    """m3d_conv2d_splitk_plan: a 512->256 3x3 on a 12x40 map (120 tiles) is split, a 96x320 map is not; a short
    workspace is refused; the split result is deterministic run to run."""
    import ctypes
    from m3dssd_amd import _hip
    from m3dssd_amd.host import standalone as S
    dev = _dev()
    L = _hip.lib()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 512, 12, 40, generator=g).to(dev)
    wt = (torch.randn(256, 512, 3, 3, generator=g) / 68.0).to(dev)
    v, _ = S._to_nhwc(x)
    out_a, ka = S.conv_nhwc(v, wt, None, None, 1, 1, cout_pad_to=64)
    out_b, kb = S.conv_nhwc(v, wt, None, None, 1, 1, cout_pad_to=64)
    assert ka[3] is not None, "expected a split-K launch"
    assert torch.equal(out_a.t, out_b.t)
    ref = F.conv2d(x, wt, None, padding=1)
    assert _relerr(S._to_nchw(out_a, 256).cpu(), ref.cpu()) < 2e-4
    d = _hip.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Cout, d.Cout_pad = 8, 96, 320, 64, 64, 64
    d.kh = d.kw = 3
    d.stride, d.pad, d.dil, d.Ho, d.Wo = 1, 1, 1, 96, 320
    splits, nbytes = ctypes.c_int(), ctypes.c_longlong()
    _hip.check(L.m3d_conv2d_splitk_plan(ctypes.byref(d), ctypes.byref(splits), ctypes.byref(nbytes)))
    assert splits.value == 1 and nbytes.value == 0
    d.H, d.W, d.Ho, d.Wo, d.Cin, d.Cout, d.Cout_pad = 12, 40, 12, 40, 512, 256, 256
    _hip.check(L.m3d_conv2d_splitk_plan(ctypes.byref(d), ctypes.byref(splits), ctypes.byref(nbytes)))
    assert splits.value >= 2 and nbytes.value == splits.value * 8 * 12 * 40 * 256 * 4
    wp, co, cop, kh, kw = S._pack(wt, 512, 64)
    out = torch.empty(8 * 12 * 40 * 256, device=dev)
    ws = torch.empty(16, device=dev)
    d.inp, d.in_cs, d.wgt, d.out, d.out_cs = v.ptr, v.cs, wp.data_ptr(), out.data_ptr(), 256
    d.sigmoid_from = -1
    d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), 64
    assert L.m3d_conv2d_forward(ctypes.byref(d), S._stream()) != 0
    assert b"workspace" in L.m3d_last_error()


def test_conv_planar_output_and_sigmoid_channels():
    """SWAP (planar NCHW) epilogue + per-channel sigmoid, as used by the head outputs / offset-mask conv."""
    import ctypes
    from m3dssd_amd import _hip
    from m3dssd_amd.host import standalone as S
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    n, ci, h, w, co = 2, 256, 16, 40, 36
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 1, 1, generator=g) / 16
    b = torch.randn(co, generator=g)
    ref = F.conv2d(x, wt, b)
    v, _ = S._to_nhwc(x.to(dev))
    wp, co_, cop, kh, kw = S._pack(wt.to(dev), v.c, 32)
    scale, shift = S._affine(co, b.to(dev), None, dev)
    out = torch.zeros(n, 3, co, h * w, device=dev)          # write into slot 1 of a [n][3][co][hw] staging tensor
    d = _hip.ConvDesc()
    d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = v.ptr, v.cs, n, h, w, v.c
    d.wgt, d.Cout, d.Cout_pad = wp.data_ptr(), co, cop
    d.kh, d.kw, d.stride, d.pad, d.dil, d.Ho, d.Wo = 1, 1, 1, 0, 1, h, w
    d.out, d.out_nchw, d.out_img_stride = out.data_ptr() + 4 * co * h * w, 1, 3 * co * h * w
    d.scale, d.shift, d.sigmoid_from = scale.data_ptr(), shift.data_ptr(), 30
    _hip.check(_hip.lib().m3d_conv2d_forward(ctypes.byref(d), S._stream()))
    got = out[:, 1].view(n, co, h, w).cpu()
    ref[:, 30:] = torch.sigmoid(ref[:, 30:])
    assert _relerr(got, ref) < 2e-4
    assert out[:, 0].abs().max().item() == 0 and out[:, 2].abs().max().item() == 0


def test_anab_nested_pooling_matches_generic_pooling():
    """m3d_anab_pool_nested (one pass, nested windows) vs m3d_anab_pool_partial + m3d_anab_pool_finish (one pass per scale) on a
    32x48 map, and vs torch adaptive_avg_pool2d of the gated features (attention.py:136-147)."""
    from m3dssd_amd import _hip
    from m3dssd_amd.engine import Engine
    dev = _dev()
    L = _hip.lib()
    B, H, W, ck, cv = 2, 32, 48, 40, 24
    C = ck + cv
    g = torch.Generator().manual_seed(11)
    kv = torch.randn(B, H, W, C, generator=g).to(dev)
    gate = torch.rand(B, H, W, 4, generator=g).to(dev)
    items, bin_scale, bin_slots, bin_inv = Engine._anab_items(H, W)
    n_bins, max_slots, keys_pad, ck_pad = len(bin_scale), int(bin_slots.max()), 352, 64
    d_items, d_bs = torch.from_numpy(items).to(dev), torch.from_numpy(bin_scale).to(dev)
    d_sl, d_inv = torch.from_numpy(bin_slots).to(dev), torch.from_numpy(bin_inv).to(dev)
    partial = torch.empty(B * n_bins * max_slots * C, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for nested in (0, 1):
        khat = torch.zeros(B, keys_pad, ck_pad, device=dev)
        vhatT = torch.zeros(B, cv, keys_pad, device=dev)
        if nested:
            scratch = torch.empty(L.m3d_anab_pool_nested_scratch_bytes(B, C) // 4, device=dev)
            _hip.check(L.m3d_anab_pool_nested(kv.data_ptr(), C, gate.data_ptr(), 4, B, H, W, ck, cv, scratch.data_ptr(),
                                              khat.data_ptr(), keys_pad, ck_pad, vhatT.data_ptr(), 0, st))
        else:
            _hip.check(L.m3d_anab_pool_partial(kv.data_ptr(), C, gate.data_ptr(), 4, d_items.data_ptr(), items.shape[0],
                                               d_bs.data_ptr(), n_bins, partial.data_ptr(), max_slots, B, H, W, C, st))
            _hip.check(L.m3d_anab_pool_finish(partial.data_ptr(), d_sl.data_ptr(), d_inv.data_ptr(), n_bins, max_slots, ck, cv,
                                              khat.data_ptr(), keys_pad, ck_pad, vhatT.data_ptr(), B, 0, st))
        outs.append((khat.cpu(), vhatT.cpu()))
    assert n_bins == 337
    x = kv.permute(0, 3, 1, 2).cpu()
    gt = gate.permute(0, 3, 1, 2).cpu()
    ref = torch.cat([F.adaptive_avg_pool2d(x * gt[:, si:si + 1], sz).flatten(2) for si, sz in enumerate((1, 4, 8, 16))], 2)
    for khat, vhatT in outs:
        assert (khat[:, :337, :ck] - ref[:, :ck].transpose(1, 2)).abs().max().item() < 2e-6
        assert (vhatT[:, :, :337] - ref[:, ck:]).abs().max().item() < 2e-6
    assert (outs[0][0] - outs[1][0]).abs().max().item() < 1e-6 and (outs[0][1] - outs[1][1]).abs().max().item() < 1e-6
    # fragment-ordered outputs (the wave-granular GEMMs' weight layout): unpack and compare with the row-major result
    kp, cvp = 384, 32                                                    # keys padded to 128, Cv a multiple of 32
    kv2 = torch.randn(B, H, W, ck + cvp, generator=g).to(dev)
    res = []
    for fr in (0, 3):
        khat = torch.zeros(B, kp * ck_pad, device=dev)
        vhatT = torch.zeros(B, cvp * kp, device=dev)
        scratch = torch.empty(L.m3d_anab_pool_nested_scratch_bytes(B, ck + cvp) // 4, device=dev)
        _hip.check(L.m3d_anab_pool_nested(kv2.data_ptr(), ck + cvp, gate.data_ptr(), 4, B, H, W, ck, cvp, scratch.data_ptr(),
                                          khat.data_ptr(), kp, ck_pad, vhatT.data_ptr(), fr, st))
        if fr:
            khat = khat.view(B, kp // 32, ck_pad // 8, 2, 32, 4).permute(0, 1, 4, 2, 3, 5).reshape(B, kp * ck_pad)
            vhatT = vhatT.view(B, cvp // 32, kp // 8, 2, 32, 4).permute(0, 1, 4, 2, 3, 5).reshape(B, cvp * kp)
        res.append((khat.cpu(), vhatT.cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert L.m3d_anab_pool_nested(kv.data_ptr(), C, gate.data_ptr(), 4, B, 24, 40, ck, cv, partial.data_ptr(),
                                  outs[0][0].data_ptr(), keys_pad, ck_pad, outs[0][1].data_ptr(), 0, st) != 0   # 24x40 does not nest


@pytest.mark.parametrize("cin,cout,cpad", [(128, 36, 64), (256, 144, 256), (128, 5, 64)])
def test_fused_head_mlp_matches_torch(cin, cout, cpad):
    """m3d_head_mlp_forward vs the unfused conv/BN/LeakyReLU chain in torch (M3d_inference_align.py:77-85)."""
    import ctypes
    from m3dssd_amd import _hip
    from m3dssd_amd.host import standalone as S
    dev = _dev()
    d, out, ref, keep = _head_case(cin + cout, cin, cout, cpad, dev)
    _hip.check(_hip.lib().m3d_head_mlp_forward(ctypes.byref(d), S._stream()))
    assert _relerr(out.view(ref.shape).cpu(), ref) < 2e-4


def test_fused_head_mlp_batched_launch():
    """m3d_head_mlp_forward_batched: five heads with their own inputs / weights / Cout in one launch, each equal to the
    torch chain and bit-identical to its single-head launch; mismatched heads are refused."""
    from m3dssd_amd import _hip
    from m3dssd_amd.host import standalone as S
    dev = _dev()
    L = _hip.lib()
    cases = [_head_case(100 + i, 128, co, 64, dev) for i, co in enumerate([36, 36, 1, 64, 17])]
    arr = (_hip.MlpDesc * len(cases))(*[c[0] for c in cases])
    _hip.check(L.m3d_head_mlp_forward_batched(arr, len(cases), S._stream()))
    torch.cuda.synchronize()
    batched = [c[1].clone() for c in cases]
    for (d, out, ref, _), got in zip(cases, batched):
        assert _relerr(got.view(ref.shape).cpu(), ref) < 2e-4
        out.zero_()
        _hip.check(L.m3d_head_mlp_forward(d, S._stream()))
        assert torch.equal(out, got)
    odd = _head_case(7, 256, 144, 256, dev)
    bad = (_hip.MlpDesc * 2)(cases[0][0], odd[0])
    assert L.m3d_head_mlp_forward_batched(bad, 2, S._stream()) != 0
    assert L.m3d_head_mlp_forward_batched(arr, 17, S._stream()) != 0


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_conv3x3_matches_torch(case, variant):
    """m3d_wino_conv3x3_forward_ex (F(2x2,3x3), fp32), LDS kernel (0) and register-resident wave kernel (1), vs F.conv2d:
    same tolerance as the direct igemm; ragged tile groups, image borders, Cout not a multiple of 32, residual."""
    import ctypes
    from m3dssd_amd import _hip
    from m3dssd_amd.host import standalone as S
    dev = _dev()
    n, ci, h, w, co, bias, bn, act, res, sg = case
    split = variant == 2                               # 2 = wave kernel in its split-K form (workspace given)
    variant = min(variant, 1)
    if variant == 1 and sg >= 0 and not split:         # the wave kernel has the sigmoid epilogue only when split: refused
        d = _hip.ConvDesc()
        d.sigmoid_from = sg
        d.Cin, d.Cout_pad, d.N, d.H, d.W = ci, 32, n, h, w
        assert _hip.lib().m3d_wino_conv3x3_variant(ctypes.byref(d)) == 0
        return
    g = torch.Generator().manual_seed(sum(case) + 7)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    b = torch.randn(co, generator=g) if bias else None
    ref = F.conv2d(x, wt, b, padding=1)
    bnm = None
    if bn:
        bnm = torch.nn.BatchNorm2d(co).eval()
        with torch.no_grad():
            bnm.weight.uniform_(0.5, 1.5, generator=g)
            bnm.bias.normal_(0, 0.2, generator=g)
            bnm.running_mean.normal_(0, 0.2, generator=g)
            bnm.running_var.uniform_(0.5, 1.5, generator=g)
        ref = bnm(ref)
    r = None
    if res:
        r = torch.randn_like(ref)
        ref = ref + r
    if sg >= 0:
        ref = ref.clone()
        ref[:, sg:] = torch.sigmoid(ref[:, sg:])
    elif act:
        ref = F.leaky_relu(ref, 0.01)
    with torch.no_grad():
        v, _ = S._to_nhwc(x.to(dev))
        rv = S._to_nhwc(r.to(dev))[0] if res else None
        out, keep = S.conv_nhwc(v, wt.to(dev), None if b is None else b.to(dev), None if bnm is None else bnm.to(dev),
                                1, 1, act=act, res=rv, sigmoid_from=sg, wino=True, wino_variant=variant, wino_splitk=split)
        got = S._to_nchw(out, co).cpu()
    assert got.shape == ref.shape
    assert _relerr(got, ref.detach()) < 2e-4


@pytest.mark.parametrize("nb", [1, 2])
@pytest.mark.parametrize("case", WINO44_CASES + [(2, 64, 24, 32, 64, True, True, 1, True), (1, 32, 8, 8, 40, True, False, 0, False)])
def test_winograd_f4x4_conv3x3_matches_torch(case, nb):
    """m3d_wino44_conv3x3_forward (Winograd F(4x4,3x3) on v_mfma_f32_16x16x4_f32, csrc/wino44_conv.hip) vs F.conv2d, same
    epilogue contract as the F(2x2,3x3) kernels.  Tolerance 2e-4 (1 + |ref|) like every fp32 conv here; the measured error is
    logged next to that of the F(2x2,3x3) wave kernel on the same operands (F(4x4) amplifies fp32 rounding ~7x)."""
    import torch.nn.functional as F
    from m3dssd_amd.host import standalone as S
    dev = _dev()
    n, ci, h, w, co, bias, bn, act, res = case
    if nb == 2 and co <= 64:
        nb = 1                                              # 64-channel layers: one 16-channel block per wave only
    g = torch.Generator().manual_seed(sum(case) + 11)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    b = torch.randn(co, generator=g) if bias else None
    ref = F.conv2d(x.double(), wt.double(), None if b is None else b.double(), padding=1)
    bnm = None
    if bn:
        bnm = torch.nn.BatchNorm2d(co).eval()
        with torch.no_grad():
            bnm.weight.uniform_(0.5, 1.5, generator=g)
            bnm.bias.normal_(0, 0.2, generator=g)
            bnm.running_mean.normal_(0, 0.2, generator=g)
            bnm.running_var.uniform_(0.5, 1.5, generator=g)
        ref = bnm.double()(ref)
        bnm = bnm.float()
    r = None
    if res:
        r = torch.randn(n, co, h, w, generator=g)
        ref = ref + r.double()
    if act:
        ref = F.leaky_relu(ref, 0.01)
    ref = ref.float()
    errs = {}
    for kind in ("wino44", "wino22"):
        if kind == "wino22" and (h % 2 or w % 2):
            continue
        with torch.no_grad():
            v, _ = S._to_nhwc(x.to(dev))
            rv = S._to_nhwc(r.to(dev))[0] if res else None
            out, keep = S.conv_nhwc(v, wt.to(dev), None if b is None else b.to(dev), None if bnm is None else bnm.to(dev), 1, 1,
                                    act=act, res=rv, wino44=(kind == "wino44"), wino44_nb=nb, wino=(kind == "wino22"), wino_variant=1)
            got = S._to_nchw(out, co).cpu()
        assert got.shape == ref.shape
        errs[kind] = ((got - ref).abs() / (1 + ref.abs())).max().item()
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_r06.jsonl"), "a") as f:
            f.write(json.dumps({"test": "wino44", "case": list(case), "nb": nb, **errs}) + "\n")
    assert errs["wino44"] < 2e-4, errs


@pytest.mark.parametrize("case,want", [((8, 256, 24, 80, 256, True, True, 1, True), 2),      # level4 at bs 8: 120 workgroups -> 240
                                       ((8, 512, 12, 40, 512, False, True, 1, True), 4),     # level5 at bs 8: 60 -> 240
                                       ((4, 128, 24, 80, 500, True, True, 0, False), 2),     # Cout 500 (pad 512), 64-channel slices
                                       ((1, 256, 8, 8, 128, True, False, 1, True), 1)])      # too small to fill the chip: no split
def test_winograd_f4x4_splitk_matches_torch(case, want):
    """Split-K form of the F(4x4,3x3) kernel (K slices as gridDim.z + m3d_launch_splitk_reduce in slice order): the plan, the
    result vs F.conv2d in fp64, and bitwise repeatability (no atomics)."""
    import ctypes
    import torch.nn.functional as F
    from m3dssd_amd import _hip
    from m3dssd_amd.host import standalone as S
    dev = _dev()
    n, ci, h, w, co, bias, bn, act, res = case
    g = torch.Generator().manual_seed(sum(case) + 5)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    b = torch.randn(co, generator=g) if bias else None
    ref = F.conv2d(x.double(), wt.double(), None if b is None else b.double(), padding=1)
    bnm = None
    if bn:
        bnm = torch.nn.BatchNorm2d(co).eval()
        with torch.no_grad():
            bnm.weight.uniform_(0.5, 1.5, generator=g)
            bnm.bias.normal_(0, 0.2, generator=g)
            bnm.running_mean.normal_(0, 0.2, generator=g)
            bnm.running_var.uniform_(0.5, 1.5, generator=g)
        ref = bnm.double()(ref)
        bnm = bnm.float()
    r = None
    if res:
        r = torch.randn(n, co, h, w, generator=g)
        ref = ref + r.double()
    if act:
        ref = F.leaky_relu(ref, 0.01)
    ref = ref.float()
    d = _hip.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Cout, d.Cout_pad = n, h, w, ci, co, -(-co // 128) * 128
    d.kh = d.kw = 3
    d.stride = d.pad = d.dil = 1
    d.Ho, d.Wo, d.in_cs, d.sigmoid_from = h, w, ci, -1
    splits, ws_bytes = ctypes.c_int(), ctypes.c_longlong()
    _hip.check(_hip.lib().m3d_wino44_splitk_plan(ctypes.byref(d), ctypes.byref(splits), ctypes.byref(ws_bytes)))
    assert splits.value == want and ws_bytes.value == (want * n * h * w * d.Cout_pad * 4 if want > 1 else 0)
    outs = []
    with torch.no_grad():
        v, _ = S._to_nhwc(x.to(dev))
        rv = S._to_nhwc(r.to(dev))[0] if res else None
        for _ in range(3):
            out, keep = S.conv_nhwc(v, wt.to(dev), None if b is None else b.to(dev), None if bnm is None else bnm.to(dev), 1, 1,
                                    act=act, res=rv, wino44=True, wino44_nb=2, wino_splitk=True)
            outs.append(S._to_nchw(out, co).cpu())
    err = ((outs[0] - ref).abs() / (1 + ref.abs())).max().item()
    assert err < 2e-4, err
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("shape", [(1, 4, 4), (2, 8, 12), (3, 20, 36), (1, 48, 160), (2, 384, 1280)])
def test_level0_winograd_f4x4_matches_torch(shape):
    """m3d_conv3x3_c16_wino (DLA level0, 3x3 16 -> 16 + folded BN + LeakyReLU as F(4x4,3x3) in three LDS phases) vs F.conv2d in
    fp64 and vs the direct kernel m3d_conv3x3_c16 on the same operands; one tile, ragged 16-tile strips, strips that cross image
    rows / images, the full 1280x384 frame."""
    import torch.nn.functional as F
    from m3dssd_amd import _hip
    from m3dssd_amd.engine import pack_wino44_c16
    L = _hip.lib()
    dev = _dev()
    n, h, w = shape
    g = torch.Generator().manual_seed(h * 7 + w)
    x = torch.randn(n, 16, h, w, generator=g)
    wt = torch.randn(16, 16, 3, 3, generator=g) / 12.0
    sc = torch.rand(16, generator=g) + 0.5
    sh = torch.randn(16, generator=g) * 0.2
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1), 0.01).float()
    xin = x.permute(0, 2, 3, 1).contiguous().to(dev)
    U = pack_wino44_c16(wt, dev)
    wd = wt.permute(2, 3, 1, 0).contiguous().to(dev)
    scd, shd = sc.to(dev), sh.to(dev)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for fn, wgt in ((L.m3d_conv3x3_c16_wino, U), (L.m3d_conv3x3_c16, wd)):
        out = torch.full((n, h, w, 16), 777.0, device=dev)
        _hip.check(fn(xin.data_ptr(), 16, wgt.data_ptr(), scd.data_ptr(), shd.data_ptr(), out.data_ptr(), 16, n, h, w, st))
        torch.cuda.synchronize()
        outs.append(out.permute(0, 3, 1, 2).cpu())
    e44 = ((outs[0] - ref).abs() / (1 + ref.abs())).max().item()
    edir = ((outs[1] - ref).abs() / (1 + ref.abs())).max().item()
    assert e44 < 2e-4 and edir < 2e-5, (e44, edir)
    again = torch.full((n, h, w, 16), 777.0, device=dev)
    _hip.check(L.m3d_conv3x3_c16_wino(xin.data_ptr(), 16, U.data_ptr(), scd.data_ptr(), shd.data_ptr(), again.data_ptr(), 16, n, h, w, st))
    torch.cuda.synchronize()
    assert torch.equal(again.permute(0, 3, 1, 2).cpu(), outs[0])


@pytest.mark.parametrize("case", [(64, 64, 24, 80, 2, False), (128, 128, 16, 48, 2, True), (256, 256, 8, 20, 1, False),
                                  (32, 64, 12, 20, 3, True)])
def test_winograd_f4x4_two_workgroups_per_cu_matches_float64(case):
    cin, cout, H, W, B, res = case
    L = _hip.lib()
    d, out, ref, keep = _w44_case(cin, cout, H, W, B, res=res)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _hip.check(L.m3d_wino44_conv3x3_forward_ex(ctypes.byref(d), 1, st))
    torch.cuda.synchronize()
    err = ((out.cpu() - ref).abs() / (1 + ref.abs())).max().item()
    assert err < 2e-4, err
    first = out.clone()
    for _ in range(10):                                       # run to run: bit-identical
        out.zero_()
        _hip.check(L.m3d_wino44_conv3x3_forward_ex(ctypes.byref(d), 1, st))
        torch.cuda.synchronize()
        assert torch.equal(out, first)


@pytest.mark.parametrize("case", [(128, 128, 16, 48, 2, True), (256, 256, 24, 80, 2, False), (512, 64, 8, 12, 1, True)])
def test_winograd_f4x4_kpair_workgroups_match_float64_and_the_plain_form(case):
    """Layers too small to fill the CU slots run K-pair workgroups (512 threads, the two halves of the input channels side by
    side, accumulators traded through LDS): against float64, against the plain 64-channel form (summation order differs:
    tolerance), and run to run bit-identical."""
    cin, cout, H, W, B, res = case
    L = _hip.lib()
    d, out, ref, keep = _w44_case(cin, cout, H, W, B, res=res)
    assert L.m3d_wino44_kpair(ctypes.byref(d)) == 1
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _hip.check(L.m3d_wino44_conv3x3_forward_ex(ctypes.byref(d), 1, st))
    torch.cuda.synchronize()
    got = out.clone()
    err = ((got.cpu() - ref).abs() / (1 + ref.abs())).max().item()
    assert err < 2e-4, err
    for _ in range(10):
        out.zero_()
        _hip.check(L.m3d_wino44_conv3x3_forward_ex(ctypes.byref(d), 1, st))
        torch.cuda.synchronize()
        assert torch.equal(out, got)
    if cout % 128 == 0:
        _hip.check(L.m3d_wino44_conv3x3_forward_ex(ctypes.byref(d), 2, st))      # the 128-channel form: one K chain
        torch.cuda.synchronize()
        assert ((out - got).abs() / (1 + got.abs())).max().item() < 1e-4


def test_winograd_f4x4_occupancy_builds_agree_bit_for_bit():
    """M3D_W44_OCC2=0 selects the one-workgroup-per-CU build of the 64-channel form: same arithmetic in the same order, so the two
    builds must produce identical bits (read once per process: a child process runs the other build)."""
    code = r'''
import ctypes, sys, torch
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from m3dssd_amd import _hip
from gpu_common import _w44_case
L = _hip.lib()
d, out, ref, keep = _w44_case(128, 128, 16, 48, 2, res=True)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_hip.check(L.m3d_wino44_conv3x3_forward_ex(ctypes.byref(d), 1, st))
torch.cuda.synchronize()
torch.save(out.cpu(), sys.argv[1])
''' % (ROOT, os.path.join(ROOT, "tests"))
    import tempfile
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for occ in ("0", "1"):
            path = os.path.join(td, "o%s.pt" % occ)
            env = dict(os.environ, M3D_W44_OCC2=occ, M3D_W44_KPAIR_MAX="0")      # (K-pair workgroups add the two K halves: not this test)
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(torch.load(path))
    assert torch.equal(outs[0], outs[1])



@pytest.mark.parametrize("c,n,h,w", [(16, 2, 5, 7), (64, 2, 6, 5), (128, 1, 5, 7), (256, 2, 3, 9), (128, 2, 24, 80), (256, 1, 12, 40)])
def test_upsample2x_add_matches_torch(c, n, h, w):
    """m3d_upsample2x_add (depthwise ConvTranspose2d(4, 2, 1) of IDAUp + the skip sum, model/pose_dla_dcn.py:261-269) against torch:
    16 to 256 channels, ragged widths, the plan's sizes, channel-sliced views with untouched neighbours."""
    from m3dssd_amd import _hip
    L, dev = _hip.lib(), _dev()
    g = torch.Generator().manual_seed(c + w)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.rand(c, 1, 4, 4, generator=g)
    skip = torch.randn(n, c, 2 * h, 2 * w, generator=g)
    ref = F.conv_transpose2d(x.double(), wt.double(), None, stride=2, padding=1, groups=c) + skip.double()
    xin = torch.full((n, h, w, c + 4), 3.0)
    xin[..., :c] = x.permute(0, 2, 3, 1)
    sk = torch.full((n, 2 * h, 2 * w, c + 8), 5.0)
    sk[..., :c] = skip.permute(0, 2, 3, 1)
    xin, sk = xin.to(dev), sk.to(dev)
    wd = wt[:, 0].permute(1, 2, 0).contiguous().to(dev)
    out = torch.full((n, 2 * h, 2 * w, c + 4), 9.0, device=dev)
    _hip.check(L.m3d_upsample2x_add(xin.data_ptr(), c + 4, wd.data_ptr(), sk.data_ptr(), c + 8, out.data_ptr(), c + 4, n, h, w, c,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    got = out[..., :c].permute(0, 3, 1, 2).cpu().double()
    assert (out[..., c:] == 9.0).all()
    assert ((got - ref).abs() <= 2e-6 * (1.0 + ref.abs())).all(), (got - ref).abs().max().item()
    out2 = torch.zeros(n, 2 * h, 2 * w, c, device=dev)
    _hip.check(L.m3d_upsample2x_add(xin.data_ptr(), c + 4, wd.data_ptr(), None, 0, out2.data_ptr(), c, n, h, w, c,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    ref2 = F.conv_transpose2d(x.double(), wt.double(), None, stride=2, padding=1, groups=c)
    assert ((out2.permute(0, 3, 1, 2).cpu().double() - ref2).abs() <= 2e-6 * (1.0 + ref2.abs())).all()


@pytest.mark.parametrize("B,h,w,keys,ck,res_mode,affine", [(2, 8, 16, 337, 168, 1, True), (1, 16, 40, 337, 168, 0, False), (3, 8, 32, 85, 64, 1, True),
                                                           (1, 48, 160, 337, 168, 1, True), (2, 8, 16, 21, 128, 0, True)])
def test_anab_attend_f32_matches_torch(B, h, w, keys, ck, res_mode, affine):
    """m3d_anab_attend_f32 (logits + softmax + P.V + residual + affine + LeakyReLU in one launch on fp32 MFMA, attention.py:207-211)
    against torch in float64: 2e-5 (1 + |ref|) -- fp32 accumulation over 168 + 337 terms, one pass over the keys with a running
    maximum.  Ragged last key tile, the padding rows / columns of khat / vhat ignored, both residual modes, no affine, slices of
    wider buffers with untouched neighbours."""
    from m3dssd_amd import _hip
    L, dev = _hip.lib(), _dev()
    g = torch.Generator().manual_seed(B * 100 + keys)
    HW, cv = h * w, 128
    kcs, kp, qcs = ck + 24, (keys + 31) // 32 * 32, ck + 8
    q = torch.randn(B * HW, ck, generator=g) * 0.5
    khat = torch.randn(B, keys, ck, generator=g) * 0.3
    vhat = torch.randn(B, cv, keys, generator=g)
    res = torch.randn(B * HW, cv, generator=g)
    scale, shift = torch.rand(cv, generator=g) + 0.5, torch.randn(cv, generator=g) * 0.1
    S = torch.einsum("bpc,bkc->bpk", q.view(B, HW, ck).double(), khat.double())
    ref = torch.einsum("bpk,bck->bpc", torch.softmax(S, dim=-1), vhat.double()).reshape(B * HW, cv)
    sc, sh = (scale.double(), shift.double()) if affine else (torch.ones(cv, dtype=torch.float64), torch.zeros(cv, dtype=torch.float64))
    ref = (ref + res.double()) * sc + sh if res_mode else ref * sc + sh + res.double()
    ref = F.leaky_relu(ref, 0.01)
    dq = torch.full((B * HW, qcs), 3.0)
    dq[:, :ck] = q
    dk = torch.full((B, kp, kcs), 7.0)                        # rows past `keys` / columns past ck must be ignored
    dk[:, :keys, :ck] = khat
    dv = torch.full((B, cv, kp), 7.0)
    dv[:, :, :keys] = vhat
    dq, dk, dv, dr = (t.contiguous().to(dev) for t in (dq, dk, dv, res))
    dsc, dsh = scale.to(dev), shift.to(dev)
    out = torch.full((B * HW, cv + 4), 512.0, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _hip.check(L.m3d_anab_attend_f32(dq.data_ptr(), qcs, dk.data_ptr(), kcs, dv.data_ptr(), B, HW, ck, keys, kp, cv, dr.data_ptr(), cv,
                                     res_mode, dsc.data_ptr() if affine else None, dsh.data_ptr() if affine else None, 1,
                                     out.data_ptr(), cv + 4, st))
    torch.cuda.synchronize()
    assert (out[:, cv:] == 512.0).all()
    got = out[:, :cv].cpu().double()
    assert ((got - ref).abs() <= 2e-5 * (1.0 + ref.abs())).all(), (got - ref).abs().max().item()
    assert L.m3d_anab_attend_f32(dq.data_ptr(), qcs, dk.data_ptr(), kcs, dv.data_ptr(), B, HW + 1, ck, keys, kp, cv, None, 0, 0, None, None, 0,
                                 out.data_ptr(), cv + 4, st) != 0
