"""GPU tests of the bf16 path (BASELINE.json configs[2]: bs = 64, bf16 storage, v_mfma_f32_32x32x16_bf16, fp32 accumulation).

Kernel level: ``m3d_conv_bf16_forward`` against torch fp32 convolutions / the CPU DCNv2 oracle evaluated on the SAME
bf16-rounded inputs and weights (products of bf16 numbers are exact in fp32, so only the accumulation order and the final
rounding differ): fp32 output modes to 2e-4 relative, bf16 outputs to one bf16 ulp of the result (2^-8 relative).
Network level: the bf16 engine against the fp32 CPU oracle with the engine's discrete decisions injected; the tolerance the
bf16 path meets is measured, logged and asserted (`BF16_BBOX3D_TOL`), next to the 1e-3 of the fp32 path.
"""
import ctypes
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from m3dssd_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BF16 = torch.bfloat16
# What the bf16 path meets against the fp32 oracle (normalised regression outputs = what the heads write, before exp() / anchor
# scaling), per output COLUMN, with the engine's discrete decisions injected into the oracle.  The typical error is bf16 rounding
# accumulated through ~50 layers; the L-inf sits at isolated pixels where the noisy centre predictions move a center_align
# sampling position (feturealign_mgpu.py:58-89) -- which is why the columns behind center_align2d (w, h) carry the largest maxima.
# Every bound = 1.3 x the value MEASURED at 1280x384 (gpurun_out/parity_r05.jsonl, tests "bf16_network" / "bf16_configs2_slice"; the
# measured values are in BF16_MEASURED for the record), so a 1.3x regression of any column fails (round 4 had one
# max bound of 2.0 for all columns against 0.77 / 0.21 measured).
BF16_COLS = {"bbox_2d": ("x", "y", "w", "h"), "bbox_3d": ("x3d", "y3d", "z3d", "w3d", "h3d", "l3d", "rY3d")}
BF16_MEASURED = {      # (max, p99.9, rms) per column at 1280x384: the LARGEST over the measurement sets of round 5 -- 2 frames
    # (seed 1234, right third zero-padded) and frames 0 / 21 / 42 / 63 of the bs-64 batch, with the round-4 and the round-5 front
    # end.  rms / p99.9 are stable to ~5 % between builds; the maxima sit at single pixels and move by +-25 % with any change of
    # the rounding points (h: 0.73 .. 0.94 over the four sets), hence "largest seen" and not one run's value.
    "bbox_2d": {"x": (0.0349, 0.0205, 0.0050), "y": (0.0299, 0.0173, 0.0046), "w": (0.620, 0.179, 0.0173), "h": (0.943, 0.197, 0.0187)},
    "bbox_3d": {"x3d": (0.0293, 0.0176, 0.0047), "y3d": (0.0324, 0.0202, 0.0051), "z3d": (0.2154, 0.0591, 0.0133),
                "w3d": (0.206, 0.0574, 0.0116), "h3d": (0.2473, 0.0570, 0.0116), "l3d": (0.2496, 0.0588, 0.0123),
                "rY3d": (0.2363, 0.0629, 0.0125)},
}
BF16_GUARD = 1.3
BF16_PROB_TOL = 0.05
BF16_PROB_MEASURED = 0.0217           # max |prob - oracle| at 1280x384
BF16_FLIP_RATE_MEASURED = 589 / 30720  # top-1 anchor decisions that differ from the free-running fp32 oracle's: 1.3 % (2 frames) .. 1.9 % (bs-64 slice)
# (an fp32 last class layer would not help: recomputing cls.6 in fp32 from the engine's bf16 hidden map leaves 182 of the 199
# flips of the 2-frame set -- they come from the ~1 % feature noise upstream, not from the last layer's rounding; logged as
# n_idx_with_fp32_cls6.  Softmax / fg_prob / top-1 already run in fp32 on fp32 logits.)
# aggregate bounds kept for the A/B-plan comparison test (two bf16 plans against each other: both sides carry the error)
BF16_BBOX_RMS_TOL = 0.03
BF16_BBOX_P999_TOL = 0.2
BF16_BBOX_MAX_TOL = 2.0


def _dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def _log(name, payload):
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_r06.jsonl"), "a") as f:
            f.write(json.dumps({"test": name, **payload}) + "\n")


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _r(t):
    """Round to bf16 and back: the value the kernel sees."""
    return t.to(BF16).float()


def _nhwc16(x, cs=None):
    """[N,C,H,W] fp32 -> bf16 NHWC device tensor with pixel stride cs (extra channels filled with a sentinel)."""
    n, c, h, w = x.shape
    cs = c if cs is None else cs
    t = torch.full((n, h, w, cs), 768.0, dtype=BF16)
    t[..., :c] = x.permute(0, 2, 3, 1).to(BF16)
    return t.contiguous().to(_dev())


def _run_conv(x, wt, bias=None, bn=None, stride=1, pad=0, act=0, res=None, res_mode=0, sigmoid_from=-1, out_mode=0, om=None,
              in_cs=None, variant=None, patch=False, wide=False, alias_res_out=False):
    """Through the C ABI.  Returns [N, Cout, Ho, Wo] fp32 (bf16 outputs widened).  alias_res_out: res = out (in-place residual);
    returns the call's status code and error text instead."""
    from m3dssd_amd import _hip
    from m3dssd_amd.engine_bf16 import pack_conv_bf16
    L = _hip.lib()
    dev = _dev()
    n, c, h, w = x.shape
    co, _, kh, kw = wt.shape
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    xin = _nhwc16(x, in_cs)
    wp, kpad = pack_conv_bf16(wt, None, None, dev)
    d = _hip.ConvBf16Desc()
    d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = xin.data_ptr(), xin.shape[3], n, h, w, c
    d.wgt, d.Cout, d.Cout_pad, d.Kpad = wp.data_ptr(), co, wp.shape[0], kpad
    d.kh, d.kw, d.stride, d.pad, d.Ho, d.Wo = kh, kw, stride, pad, ho, wo
    keep = [xin, wp]
    scale = torch.ones(co)
    shift = torch.zeros(co) if bias is None else bias.clone()
    if bn is not None:
        g, b, m, v = bn
        s = g / torch.sqrt(v + 1e-5)
        shift = (shift - m) * s + b
        scale = s
    if bias is not None or bn is not None:
        sc, sh = scale.to(dev).contiguous(), shift.to(dev).contiguous()
        d.scale, d.shift = sc.data_ptr(), sh.data_ptr()
        keep += [sc, sh]
    if res is not None:
        r = _nhwc16(res)
        d.res, d.res_cs, d.res_mode = r.data_ptr(), r.shape[3], res_mode
        keep.append(r)
    d.act, d.sigmoid_from, d.groups = act, sigmoid_from, 1
    if wide:                # 128 x 128 wave-tile kernel: the same weights in fragment order (engine_bf16.PackedBf16.wave3x3)
        cop = wp.shape[0]
        wv = wp[:, :9 * c].reshape(cop // 128, 4, 32, 9, c // 32, 2, 2, 8).permute(0, 4, 3, 5, 1, 6, 2, 7).contiguous()
        d.wgt_wave = wv.data_ptr()
        keep.append(wv)
    if om is not None:
        o = om.to(dev).contiguous()
        d.dcn_offmask, d.dcn_om_cs = o.data_ptr(), o.shape[-1]
        keep.append(o)
        if patch:           # LDS-patch DCNv2 kernel: fp16 weight copy + the device scratch of the |offset| bound
            nws = max(256, _hip.lib().m3d_conv_bf16_dcn_ws_bytes(n, ho, wo) // 4)
            w16, ws = wp.float().to(torch.float16).contiguous(), torch.full((nws,), 7, device=dev, dtype=torch.int32)
            d.wgt_f16, d.dcn_ws, d.dcn_ws_bytes = w16.data_ptr(), ws.data_ptr(), 4 * nws
            keep += [w16, ws]
    if out_mode == 0:
        ocs = (co + 7) // 8 * 8 + 8
        out = torch.full((n, ho, wo, ocs), 512.0, device=dev, dtype=BF16)
        d.out, d.out_cs = out.data_ptr(), ocs
    elif out_mode == 1:
        ocs = (co + 3) // 4 * 4 + 4
        out = torch.full((n, ho, wo, ocs), 512.0, device=dev, dtype=torch.float32)
        d.out, d.out_cs = out.data_ptr(), ocs
    else:
        out = torch.full((n, co + 1, ho * wo), 512.0, device=dev, dtype=torch.float32)
        d.out, d.out_img_stride = out.data_ptr(), (co + 1) * ho * wo
    d.out_mode = out_mode
    if alias_res_out:
        d.res, d.res_cs, d.res_mode = d.out, d.out_cs, 0
        rc = L.m3d_conv_bf16_forward(ctypes.byref(d), _st())
        torch.cuda.synchronize()
        return rc, L.m3d_last_error().decode()
    if variant is not None:
        assert L.m3d_conv_bf16_variant(ctypes.byref(d)) == variant
    _hip.check(L.m3d_conv_bf16_forward(ctypes.byref(d), _st()))
    torch.cuda.synchronize()
    if out_mode == 2:
        assert (out[:, co] == 512.0).all()                     # the channel past Cout is untouched
        return out[:, :co].view(n, co, ho, wo).cpu()
    assert (out[..., co:].float() == 512.0).all()              # nothing is written past Cout
    return out[..., :co].float().permute(0, 3, 1, 2).contiguous().cpu()


def _check(got, ref, out_mode):
    scale = ref.abs().max().item() + 1e-6
    if out_mode == 0:        # one bf16 rounding of the result (+ fp32 accumulation noise)
        tol = 2.0 ** -8 * ref.abs() + 1e-3 * scale
    else:
        tol = 2e-4 * ref.abs() + 2e-4 * scale
    bad = (got - ref).abs() > tol
    assert not bad.any(), ((got - ref).abs().max().item(), scale, int(bad.sum()))


CONV_CASES = [
    # n, c, h, w, co, k, stride, pad, act, res, sigmoid_from, out_mode
    (2, 16, 24, 40, 16, 3, 1, 1, 1, False, -1, 0),        # level0: Cin 16, K = 144 -> 192 (zero-padded K), Cout 16 < 32
    (2, 16, 24, 40, 32, 3, 2, 1, 1, False, -1, 0),        # level1: stride 2
    (1, 32, 17, 23, 64, 3, 2, 1, 1, False, -1, 0),        # odd sizes, M tail
    (2, 64, 12, 20, 64, 3, 1, 1, 1, True, -1, 0),         # residual block conv2
    (1, 128, 16, 40, 128, 3, 1, 1, 1, True, -1, 0),
    (1, 448, 12, 20, 128, 1, 1, 0, 1, False, -1, 0),      # tree root: Cin not a power of two (1x1)
    (1, 256, 6, 10, 512, 3, 2, 1, 1, False, -1, 0),
    (2, 128, 16, 40, 27, 3, 1, 1, 0, False, 18, 1),       # offset / mask conv: fp32 NHWC out, sigmoid on the mask channels
    (2, 256, 8, 20, 144, 1, 1, 0, 0, False, -1, 2),       # cls.6: planar fp32 out, Cout_pad 192
    (1, 256, 8, 20, 36, 1, 1, 0, 0, False, -1, 2),
    (1, 128, 16, 40, 300, 1, 1, 0, 0, False, 296, 1),     # ANAB K|V|S conv: Cout 300 (pad 320), fp32 out
    (1, 128, 16, 40, 168, 1, 1, 0, 0, False, -1, 0),      # ANAB Q conv: Cout 168, partial last 8-channel store
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_bf16_matches_torch(case):
    n, c, h, w, co, k, stride, pad, act, use_res, sg, om = case
    g = torch.Generator().manual_seed(sum(case) + 3)
    x = _r(torch.randn(n, c, h, w, generator=g))
    wt = _r(torch.randn(co, c, k, k, generator=g) / (c * k * k) ** 0.5)
    bias = torch.randn(co, generator=g) * 0.1
    bn = (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1,
          torch.rand(co, generator=g) + 0.5)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = _r(torch.randn(n, co, ho, wo, generator=g)) if use_res else None
    ref = F.batch_norm(F.conv2d(x, wt, bias, stride=stride, padding=pad), bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    if res is not None:
        ref = ref + res
    if sg >= 0:
        ref = torch.cat([F.leaky_relu(ref[:, :sg], 0.01) if act else ref[:, :sg], torch.sigmoid(ref[:, sg:])], 1)
    elif act:
        ref = F.leaky_relu(ref, 0.01)
    got = _run_conv(x, wt, bias, bn, stride, pad, act, res, 0, sg, om, in_cs=c + 8 if c % 16 == 0 else None)
    assert got.shape == ref.shape
    _check(got, ref, om)


HALO_CASES = [
    # n, c, h, w, co, act, res, sigmoid_from, out_mode, kernel variant (m3d_conv_bf16_variant)
    (2, 64, 15, 31, 64, 1, True, -1, 0, 1),          # 8 x 16 patches, ragged right / bottom patches, BN 64
    (1, 128, 23, 47, 128, 1, True, -1, 0, 1),        # odd map, BN 128
    (2, 64, 12, 20, 64, 1, True, -1, 0, 0),          # 12 x 20 map: patches 47 % full -> implicit-GEMM tile
    (3, 64, 8, 16, 27, 0, False, 18, 1, 1),          # offset / mask conv: Cout_pad 32 (every wave takes all channels), fp32 out
    (2, 128, 21, 45, 27, 0, False, 18, 1, 1),        # the same on a ragged map, two chunks
    (1, 256, 12, 40, 256, 1, False, -1, 0, 0),       # 12 x 40 map: patches would be 62 % full -> implicit-GEMM tile
    (2, 128, 16, 48, 192, 0, False, -1, 2, 1),       # planar fp32 output, Cout 192 (BN 64, 3 channel tiles)
    (56, 64, 22, 62, 512, 1, True, -1, 0, 1),        # 128-channel tiles: 8 x 16 patches, 32-channel weight half-steps, ragged, 4 channel tiles
    (128, 64, 24, 96, 64, 1, False, -1, 1, 2),       # 8 x 32 patches, BN 64, fp32 NHWC output
    (256, 128, 16, 64, 64, 1, True, -1, 0, 2),       # 8 x 32 patches, two 64-channel chunks: the patch buffer is refilled mid-loop
    (16, 256, 16, 32, 256, 1, True, -1, 0, 1),       # four chunks with half-steps, two channel tiles
]


@pytest.mark.parametrize("case", HALO_CASES)
def test_conv_bf16_halo_tile_matches_torch(case):
    """3x3 / stride 1 / pad 1: the halo-tile kernels (input patch resident in LDS, K order (chunk, tap)) against torch on the
    bf16-rounded operands; the variant the library picks is pinned so a heuristic change cannot silently drop the coverage."""
    n, c, h, w, co, act, use_res, sg, om, variant = case
    g = torch.Generator().manual_seed(sum(case) + 5)
    x = _r(torch.randn(n, c, h, w, generator=g))
    wt = _r(torch.randn(co, c, 3, 3, generator=g) / (c * 9) ** 0.5)
    bias = torch.randn(co, generator=g) * 0.1
    res = _r(torch.randn(n, co, h, w, generator=g)) if use_res else None
    ref = F.conv2d(x, wt, bias, padding=1)
    if res is not None:
        ref = ref + res
    if sg >= 0:
        ref = torch.cat([F.leaky_relu(ref[:, :sg], 0.01) if act else ref[:, :sg], torch.sigmoid(ref[:, sg:])], 1)
    elif act:
        ref = F.leaky_relu(ref, 0.01)
    got = _run_conv(x, wt, bias, None, 1, 1, act, res, 0, sg, om, in_cs=c + 8, variant=variant)
    assert got.shape == ref.shape
    _check(got, ref, om)


WIDE_CASES = [
    # n, c, h, w, co, act, res, bn
    (1, 64, 8, 16, 128, 0, False, False),            # one patch, two chunks (the minimum): every patch border is an image border
    (2, 64, 16, 32, 128, 1, True, True),             # 2 x 2 patches per image, two chunks, residual + BN + LeakyReLU
    (3, 128, 24, 48, 256, 1, True, True),            # two channel groups, four chunks, 27 patches (ragged last workgroup)
    (1, 64, 8, 48, 100, 1, False, True),             # Cout 100 (pad 128)
    (5, 256, 24, 80, 256, 1, True, True),            # level4 geometry
    (2, 128, 48, 160, 128, 1, True, True),           # level3 geometry
]


@pytest.mark.parametrize("case", WIDE_CASES)
def test_conv_bf16_wide_tile_matches_torch(case):
    """3x3 / stride 1 / pad 1 on the 128-pixel x 128-channel wave-tile kernel (csrc/bf16_conv_wide.hip: per-wave input patch in LDS,
    weights global -> register in fragment order, K order (chunk of 32, tap, 16)) against torch on the bf16-rounded operands, and
    bitwise against a second launch."""
    n, c, h, w, co, act, use_res, use_bn = case
    g = torch.Generator().manual_seed(sum(case) + 9)
    x = _r(torch.randn(n, c, h, w, generator=g))
    wt = _r(torch.randn(co, c, 3, 3, generator=g) / (c * 9) ** 0.5)
    bias = torch.randn(co, generator=g) * 0.1
    bn = (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1,
          torch.rand(co, generator=g) + 0.5) if use_bn else None
    res = _r(torch.randn(n, co, h, w, generator=g)) if use_res else None
    ref = F.conv2d(x, wt, bias, padding=1)
    if bn is not None:
        ref = F.batch_norm(ref, bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    if res is not None:
        ref = ref + res
    if act:
        ref = F.leaky_relu(ref, 0.01)
    got = _run_conv(x, wt, bias, bn, 1, 1, act, res, 0, -1, 0, in_cs=c + 8, variant=5, wide=True)
    assert got.shape == ref.shape
    _check(got, ref, 0)
    again = _run_conv(x, wt, bias, bn, 1, 1, act, res, 0, -1, 0, in_cs=c + 8, variant=5, wide=True)
    assert torch.equal(got, again)
    # res_mode 1 ((acc + res) * scale + shift): the other place the residual enters
    if use_res and bn is not None:
        s = bn[0] / torch.sqrt(bn[3] + 1e-5)
        ref1 = (F.conv2d(x, wt, None, padding=1) + res) * s.view(1, -1, 1, 1) + ((bias - bn[2]) * s + bn[1]).view(1, -1, 1, 1)
        if act:
            ref1 = F.leaky_relu(ref1, 0.01)
        got1 = _run_conv(x, wt, bias, bn, 1, 1, act, res, 1, -1, 0, in_cs=c + 8, variant=5, wide=True)
        _check(got1, ref1, 0)


@pytest.mark.parametrize("n,h,w,act,use_res,res_mode", [(8, 128, 256, 1, True, 0), (9, 96, 320, 1, False, 0), (17, 64, 256, 0, True, 1)])
def test_conv_bf16_c64_persistent_kernel_matches_torch_and_the_halo_tile(n, h, w, act, use_res, res_mode):
    """csrc/bf16_conv_c64.hip (3x3 64 -> 64 on persistent workgroups, all weights resident in LDS, next tile's halo in flight; DLA
    level2) against torch on the bf16-rounded operands, run to run, and batch-invariant (a sub-batch = the same bits per image: one
    kernel for every batch size).  Tile counts that do not divide by the workgroups
    (1 080 = 9 x 12 x 10; 1 088 = 17 x 8 x 8) and exactly four tiles per workgroup (1 024)."""
    c = co = 64
    g = torch.Generator().manual_seed(n + h)
    x = _r(torch.randn(n, c, h, w, generator=g))
    wt = _r(torch.randn(co, c, 3, 3, generator=g) / (c * 9) ** 0.5)
    bias = torch.randn(co, generator=g) * 0.1
    bn = (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1, torch.rand(co, generator=g) + 0.5)
    res = _r(torch.randn(n, co, h, w, generator=g)) if use_res else None
    sc = bn[0] / torch.sqrt(bn[3] + 1e-5)
    conv = F.conv2d(x, wt, None, padding=1)
    aff = lambda t: t * sc.view(1, -1, 1, 1) + ((bias - bn[2]) * sc + bn[1]).view(1, -1, 1, 1)     # noqa: E731
    ref = aff(conv + res) if (use_res and res_mode == 1) else (aff(conv) + (res if use_res else 0.0))
    if act:
        ref = F.leaky_relu(ref, 0.01)
    got = _run_conv(x, wt, bias, bn, 1, 1, act, res, res_mode, -1, 0, in_cs=c + 8, variant=8)
    _check(got, ref, 0)
    again = _run_conv(x, wt, bias, bn, 1, 1, act, res, res_mode, -1, 0, in_cs=c + 8, variant=8)
    assert torch.equal(got, again)
    k = 3                                                        # batch invariance: a sub-batch runs the same kernel, same bits per image
    sub = _run_conv(x[:k], wt, bias, bn, 1, 1, act, None if res is None else res[:k], res_mode, -1, 0, in_cs=c + 8, variant=8)
    assert torch.equal(got[:k], sub)
    # the halo-tile kernel it replaces (M3D_BF16_C64=0) sums the taps in another order: equal to fp32 summation noise, i.e. one bf16 ulp
    # of the result at a few entries -- checked against torch above, like every other conv kernel


def test_dcn_bf16_run_to_run_identical():
    """Deformable mode at full-size grids (3840 workgroups, two waves per SIMD): 24 launches on the same operands must agree bit
    for bit.  Sampling code written with compares (v_cmp -> s_and_b64 -> v_cndmask on SGPR lane masks) dropped one corner in
    lanes 48-63 of a wave once per 10^5..10^6 (pixel, tap) states, at a rate that changed with every recompile (0 of 800
    launches for one build, every launch for another): csrc/common.h `dcn_corners` keeps every decision in VGPR sign masks.
    tools/dcn_determinism.py is the long form of this test (300 launches, per-corner analysis of a difference)."""
    from m3dssd_amd import _hip
    from m3dssd_amd.engine_bf16 import pack_conv_bf16
    L, dev = _hip.lib(), _dev()
    for cin, cout, h, w, b, k in [(128, 128, 48, 160, 64, 3), (256, 256, 24, 80, 64, 3), (128, 128, 48, 160, 64, 1)]:
        g = torch.Generator().manual_seed(cin + k)
        x = torch.randn(b * h * w, cin, generator=g).to(BF16).to(dev)
        wp, kpad = pack_conv_bf16(torch.randn(cout, cin, k, k, generator=g) / (k * k * cin) ** 0.5, None, None, dev)
        kk = k * k
        om = torch.cat([torch.randn(b * h * w, 2 * kk, generator=g) * 2.0, torch.rand(b * h * w, kk, generator=g),
                        torch.zeros(b * h * w, 32 - 3 * kk)], 1).contiguous().to(dev)
        outs = []
        for _ in range(24):
            out = torch.zeros(b * h * w, cout, device=dev, dtype=BF16)
            d = _hip.ConvBf16Desc()
            d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.data_ptr(), cin, b, h, w, cin
            d.wgt, d.Cout, d.Cout_pad, d.Kpad = wp.data_ptr(), cout, wp.shape[0], kpad
            d.kh = d.kw = k
            d.stride, d.pad, d.Ho, d.Wo = 1, k // 2, h, w
            d.out, d.out_cs, d.out_mode, d.act, d.sigmoid_from, d.groups = out.data_ptr(), cout, 0, 1, -1, 1
            d.dcn_offmask, d.dcn_om_cs = om.data_ptr(), 32
            _hip.check(L.m3d_conv_bf16_forward(ctypes.byref(d), _st()))
            torch.cuda.synchronize()
            outs.append(out)
        assert torch.isfinite(outs[0].float()).all()
        for o in outs[1:]:
            assert torch.equal(outs[0], o), (cin, k, int((outs[0] != o).sum()))


def test_conv_bf16_residual_before_affine_and_argument_checks():
    """res_mode 1 (ANAB: BN after the residual add) and the C-ABI argument validation."""
    from m3dssd_amd import _hip
    L = _hip.lib()
    g = torch.Generator().manual_seed(11)
    x = _r(torch.randn(1, 64, 8, 16, generator=g))
    wt = _r(torch.randn(128, 64, 1, 1, generator=g) / 8)
    res = _r(torch.randn(1, 128, 8, 16, generator=g))
    bn = (torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1, torch.randn(128, generator=g) * 0.1,
          torch.rand(128, generator=g) + 0.5)
    ref = F.leaky_relu(F.batch_norm(F.conv2d(x, wt) + res, bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5), 0.01)
    got = _run_conv(x, wt, None, bn, 1, 0, 1, res, 1)
    _check(got, ref, 0)
    d = _hip.ConvBf16Desc()
    assert L.m3d_conv_bf16_forward(ctypes.byref(d), _st()) == -1            # null pointers
    t = torch.zeros(64, device=_dev())
    d.inp = d.wgt = d.out = t.data_ptr()
    d.N, d.H, d.W, d.Cin, d.in_cs, d.Cout, d.Cout_pad, d.Kpad = 1, 4, 4, 24, 24, 32, 32, 256
    d.kh = d.kw = 3
    d.stride, d.pad, d.Ho, d.Wo, d.groups, d.out_cs = 1, 1, 4, 4, 1, 32
    assert L.m3d_conv_bf16_forward(ctypes.byref(d), _st()) == -1            # 3x3 with Cin = 24 (not a power of two)
    assert b"power-of-two" in L.m3d_last_error()


def test_conv_bf16_grouped_and_per_image_weights():
    """groups (the RPN heads of one feature map in one launch) and per-image weights (ANAB logits / P.V GEMMs)."""
    from m3dssd_amd import _hip
    from m3dssd_amd.engine_bf16 import pack_conv_bf16
    L = _hip.lib()
    dev = _dev()
    g = torch.Generator().manual_seed(4)
    G, n, h, w, ci, co = 3, 2, 8, 16, 64, 36
    x = _r(torch.randn(n, G * ci, h, w, generator=g))
    wts = [_r(torch.randn(co, ci, 1, 1, generator=g) / 8) for _ in range(G)]
    sc = torch.rand(G, co, generator=g) + 0.5
    sh = torch.randn(G, co, generator=g) * 0.1
    xin = _nhwc16(x)
    wp = torch.cat([pack_conv_bf16(wt, 64, None, dev)[0] for wt in wts], 0).contiguous()
    out = torch.zeros(n, G * co, h * w, device=dev)
    d = _hip.ConvBf16Desc()
    d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = xin.data_ptr(), G * ci, n, h, w, ci
    d.wgt, d.Cout, d.Cout_pad, d.Kpad = wp.data_ptr(), co, 64, 64
    d.kh = d.kw = d.stride = 1
    d.Ho, d.Wo, d.out, d.out_mode, d.out_img_stride = h, w, out.data_ptr(), 2, G * co * h * w
    scd, shd = sc.to(dev).contiguous(), sh.to(dev).contiguous()
    d.scale, d.shift, d.sigmoid_from = scd.data_ptr(), shd.data_ptr(), -1
    d.groups, d.in_group_off, d.wgt_group_off, d.out_group_off, d.ss_group_off = G, ci, 64 * 64, co * h * w, co
    _hip.check(L.m3d_conv_bf16_forward(ctypes.byref(d), _st()))
    torch.cuda.synchronize()
    got = out.view(n, G, co, h, w).cpu()
    for gi in range(G):
        ref = F.conv2d(x[:, gi * ci:(gi + 1) * ci], wts[gi]) * sc[gi].view(1, -1, 1, 1) + sh[gi].view(1, -1, 1, 1)
        _check(got[:, gi], ref, 2)
    # per-image weights: image b multiplies by its own [Cout][K] matrix (Ho*Wo = 128 pixels per image)
    n, h, w, ci, co = 3, 8, 16, 192, 337
    x = _r(torch.randn(n, ci, h, w, generator=g))
    wimg = _r(torch.randn(n, co, ci, generator=g) / 14)
    cop = 384
    wp = torch.zeros(n, cop, ci, dtype=BF16)
    wp[:, :co] = wimg.to(BF16)
    wp = wp.to(dev).contiguous()
    xin = _nhwc16(x)
    out = torch.zeros(n, h, w, cop, device=dev)
    d = _hip.ConvBf16Desc()
    d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = xin.data_ptr(), ci, n, h, w, ci
    d.wgt, d.wgt_img_stride, d.Cout, d.Cout_pad, d.Kpad = wp.data_ptr(), cop * ci, co, cop, ci
    d.kh = d.kw = d.stride = 1
    d.Ho, d.Wo, d.out, d.out_cs, d.out_mode, d.groups, d.sigmoid_from = h, w, out.data_ptr(), cop, 1, 1, -1
    _hip.check(L.m3d_conv_bf16_forward(ctypes.byref(d), _st()))
    torch.cuda.synchronize()
    got = out[..., :co].permute(0, 3, 1, 2).cpu()
    ref = torch.einsum("nok,nkhw->nohw", wimg, x)
    _check(got, ref, 1)


@pytest.mark.parametrize("form", [1, 2])
@pytest.mark.parametrize("G,n,h,w,cout", [(1, 2, 8, 16, 36), (4, 1, 13, 21, 36), (2, 2, 16, 40, 5), (6, 3, 48, 160, 36), (1, 1, 4, 8, 64)])
def test_fused_head_mlp_bf16_matches_torch_chain(G, n, h, w, cout, form):
    """m3d_head_mlp_bf16_forward / m3d_head_mlp2_bf16_forward (3 layers, hidden activations in LDS, heads of one map in one
    launch) against the torch chain on the same bf16-rounded input / weights with the hidden activations rounded to bf16 where the
    first form rounds them (the second keeps them in fp16 and folds the scales into the weights: inside the same bound).  Ragged
    tiles (13 x 21), several tiles per workgroup and persistent workgroups per head (6 heads x 3 x 48 x 160), Cout = 64."""
    from m3dssd_amd import _hip
    from m3dssd_amd.engine_bf16 import pack_head2
    L = _hip.lib()
    dev = _dev()
    g = torch.Generator().manual_seed(G * 100 + h)
    x = _r(torch.randn(n, 128, h, w, generator=g))
    M, HW = n * h * w, h * w
    xin = _nhwc16(x, 136)
    w1 = _r(torch.randn(G, 256, 128, generator=g) / 128 ** 0.5)
    w2 = _r(torch.randn(G, 256, 256, generator=g) / 16)
    w3 = torch.zeros(G, 64, 256)
    w3[:, :cout] = _r(torch.randn(G, cout, 256, generator=g) / 16)
    aff = [torch.rand(G, c, generator=g) + 0.5 for c in (256, 256, cout)]
    sh = [torch.randn(G, c, generator=g) * 0.1 for c in (256, 256, cout)]
    out = torch.full((n, G * cout + 1, HW), 512.0, device=dev)
    dv = [t.to(dev).contiguous() for t in (w1.to(BF16), w2.to(BF16), w3.to(BF16), aff[0], sh[0], aff[1], sh[1], aff[2], sh[2])]
    if form == 1:
        d = _hip.HeadBf16Desc()
        d.inp, d.in_cs, d.M, d.Cin = xin.data_ptr(), 136, M, 128
        d.w1, d.w2, d.w3, d.s1, d.t1, d.s2, d.t2, d.s3, d.t3 = (t.data_ptr() for t in dv)
        d.Cout, d.Cout_pad, d.out = cout, 64, out.data_ptr()
        d.out_group_off, d.out_img_stride, d.HW, d.groups = cout * HW, (G * cout + 1) * HW, HW, G
        _hip.check(L.m3d_head_mlp_bf16_forward(ctypes.byref(d), _st()))
    else:
        pk = pack_head2([(w1[gi], aff[0][gi], sh[0][gi], w2[gi], aff[1][gi], sh[1][gi], w3[gi, :cout], aff[2][gi], sh[2][gi])
                         for gi in range(G)], dev)
        d = _hip.Head2Bf16Desc()
        d.inp, d.in_cs, d.M = xin.data_ptr(), 136, M
        d.w1f, d.w2f, d.w3, d.t1, d.t2, d.t3 = (t.data_ptr() for t in pk)
        d.Cout, d.out = cout, out.data_ptr()
        d.out_group_off, d.out_img_stride, d.HW, d.groups = cout * HW, (G * cout + 1) * HW, HW, G
        _hip.check(L.m3d_head_mlp2_bf16_forward(ctypes.byref(d), _st()))
    torch.cuda.synchronize()
    got = out.cpu()
    assert (got[:, G * cout] == 512.0).all()
    xf = x.permute(0, 2, 3, 1).reshape(M, 128)
    for gi in range(G):
        h1 = _r(F.leaky_relu(xf @ w1[gi].T * aff[0][gi] + sh[0][gi], 0.01))
        h2 = _r(F.leaky_relu(h1 @ w2[gi].T * aff[1][gi] + sh[1][gi], 0.01))
        ref = (h2 @ w3[gi, :cout].T * aff[2][gi] + sh[2][gi]).view(n, HW, cout).permute(0, 2, 1)
        e = (got[:, gi * cout:(gi + 1) * cout] - ref).abs().max().item()
        assert e < 4e-3 * (1.0 + ref.abs().max().item()), (gi, e)      # a hidden value may round to the neighbouring bf16


def test_fused_head2_saturates_fp16_hidden_activations_instead_of_going_nan():
    """the round-5 head keeps its hidden activations in fp16.  Inputs that drive a hidden value past +-65504 (bf16
    activations of magnitude ~3e3 against unit-scale weights: |h1| up to ~1e6) must not turn into Inf / NaN in the planar outputs:
    the epilogues clamp the packed pair to the finite fp16 range (M3D_F16_SATURATE), i.e. the head degrades like a saturating
    quantiser.  In-range pixels of the same launch stay within the usual bound."""
    from m3dssd_amd import _hip
    from m3dssd_amd.engine_bf16 import pack_head2
    L, dev = _hip.lib(), _dev()
    g = torch.Generator().manual_seed(5)
    n, h, w, cout, G = 1, 8, 16, 36, 2
    M, HW = n * h * w, h * w
    x = _r(torch.randn(n, 128, h, w, generator=g))
    x[:, :, :4] *= 30000.0                                      # upper half of the map: far out of the fp16 range after layer 1
    x = _r(x)
    xin = _nhwc16(x, 136)
    w1 = _r(torch.randn(G, 256, 128, generator=g) / 128 ** 0.5 * 4)
    w2 = _r(torch.randn(G, 256, 256, generator=g) / 16)
    w3 = _r(torch.randn(G, cout, 256, generator=g) / 16)
    one, zero = torch.ones(256), torch.zeros(256)
    pk = pack_head2([(w1[gi], one, zero, w2[gi], one, zero, w3[gi], torch.ones(cout), torch.zeros(cout)) for gi in range(G)], dev)
    out = torch.full((n, G * cout, HW), 512.0, device=dev)
    d = _hip.Head2Bf16Desc()
    d.inp, d.in_cs, d.M = xin.data_ptr(), 136, M
    d.w1f, d.w2f, d.w3, d.t1, d.t2, d.t3 = (t.data_ptr() for t in pk)
    d.Cout, d.out = cout, out.data_ptr()
    d.out_group_off, d.out_img_stride, d.HW, d.groups = cout * HW, G * cout * HW, HW, G
    _hip.check(L.m3d_head_mlp2_bf16_forward(ctypes.byref(d), _st()))
    torch.cuda.synchronize()
    got = out.cpu()
    xf = x.permute(0, 2, 3, 1).reshape(M, 128)
    h1 = F.leaky_relu(xf @ w1[0].T, 0.01)
    assert float(h1[:4 * w].abs().max()) > 65504.0 and float(h1[4 * w:].abs().max()) < 6e4      # the test does overflow fp16
    assert torch.isfinite(got).all()
    lo = slice(4 * w, HW)                                        # the in-range half: the usual bound
    for gi in range(G):
        a = _r(F.leaky_relu(xf @ w1[gi].T, 0.01))
        b = _r(F.leaky_relu(a @ w2[gi].T, 0.01))
        ref = (b @ w3[gi].T).view(n, HW, cout).permute(0, 2, 1)
        e = (got[:, gi * cout:(gi + 1) * cout, lo] - ref[:, :, lo]).abs().max().item()
        assert e < 4e-3 * (1.0 + ref[:, :, lo].abs().max().item()), (gi, e)


@pytest.mark.parametrize("n,cin,H,W,with_bottom", [(2, 32, 32, 64, False), (1, 64, 48, 96, True), (2, 128, 16, 32, True), (1, 256, 24, 80, True),
                                                   (3, 32, 192, 640, False), (1, 64, 20, 36, True)])
def test_tree_entry_bf16_matches_torch(n, cin, H, W, with_bottom):
    """m3d_tree_entry_bf16_forward (max-pool + 1x1 project + 3x3 stride-2 conv1 of a DLA tree in one launch) against torch on the
    bf16-rounded input / weights: bottom bit-exact (a max of bf16 values), t / res to one bf16 ulp of the result (the kernel folds
    the scales into fp16 weights: + 2^-9 relative).  Every level's channel pair, image borders, partial tiles (24 x 80 -> 12 x 40,
    20 x 36 -> 10 x 18), channel slices of a wider buffer, the full-size level2 map."""
    from m3dssd_amd import _hip
    from m3dssd_amd.engine_bf16 import pack_tree_entry
    L, dev = _hip.lib(), _dev()
    co = 2 * cin
    g = torch.Generator().manual_seed(cin + H)
    x = _r(torch.randn(n, cin, H, W, generator=g))
    w1 = _r(torch.randn(co, cin, 3, 3, generator=g) / (9 * cin) ** 0.5)
    wp = _r(torch.randn(co, cin, 1, 1, generator=g) / cin ** 0.5)
    s1, t1 = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1
    sp, tp = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1
    xin = _nhwc16(x, cin + 8)
    Ho, Wo = H // 2, W // 2
    t = torch.full((n, Ho, Wo, co + 8), 512.0, device=dev, dtype=BF16)
    res = torch.full((n, Ho, Wo, co), 512.0, device=dev, dtype=BF16)
    bot = torch.full((n, Ho, Wo, cin + 16), 512.0, device=dev, dtype=BF16)
    wf = pack_tree_entry(w1, s1, wp, sp, dev)
    dv = [v.to(dev).contiguous() for v in (t1, tp)]
    d = _hip.TreeEntryBf16Desc()
    d.inp, d.in_cs, d.N, d.H, d.W, d.Cin, d.Cout = xin.data_ptr(), cin + 8, n, H, W, cin, co
    d.wfrag, d.shift1, d.shiftp = wf.data_ptr(), dv[0].data_ptr(), dv[1].data_ptr()
    d.t, d.t_cs, d.res, d.res_cs = t.data_ptr(), co + 8, res.data_ptr(), co
    if with_bottom:
        d.bottom, d.bottom_cs = bot.data_ptr() + 2 * 8, cin + 16          # a channel slice [8, 8 + cin) of a wider buffer
    assert L.m3d_tree_entry_bf16_applicable(ctypes.byref(d)) == 1
    _hip.check(L.m3d_tree_entry_bf16_forward(ctypes.byref(d), _st()))
    torch.cuda.synchronize()
    pooled = F.max_pool2d(x, 2, 2)
    ref_t = F.leaky_relu(F.conv2d(x, w1, None, stride=2, padding=1) * s1.view(1, -1, 1, 1) + t1.view(1, -1, 1, 1), 0.01)
    ref_r = F.conv2d(pooled, wp) * sp.view(1, -1, 1, 1) + tp.view(1, -1, 1, 1)
    assert (t[..., co:].float() == 512.0).all()
    for name, got, ref in (("t", t[..., :co], ref_t), ("res", res, ref_r)):
        gotf = got.float().permute(0, 3, 1, 2).cpu()
        err = (gotf - ref).abs()
        tol = 2.0 ** -7 * ref.abs() + 2e-3 * ref.abs().max()
        assert (err <= tol).all(), (name, err.max().item(), ref.abs().max().item(), int((err > tol).sum()))
    if with_bottom:
        assert torch.equal(bot[..., 8:8 + cin].float().permute(0, 3, 1, 2).cpu(), pooled)
        assert (bot[..., :8].float() == 512.0).all() and (bot[..., 8 + cin:].float() == 512.0).all()
    else:
        assert (bot.float() == 512.0).all()


@pytest.mark.parametrize("n,h,w", [(2, 8, 16), (1, 13, 21), (3, 48, 160)])
def test_anab_qkvs_bf16_matches_torch(n, h, w):
    """m3d_anab_qkvs_bf16_forward (query | key | value | gates of ANAB in one launch) against torch on the bf16-rounded operands: q and
    k|v to one bf16 ulp, the padding rows of q exact zeros, gates = sigmoid in fp32 to 2e-4; ragged tiles, untouched neighbours."""
    from m3dssd_amd import _hip
    from m3dssd_amd.engine_bf16 import _head2_frag
    L, dev = _hip.lib(), _dev()
    ck, cv, ns, ckp = 168, 128, 4, 192
    g = torch.Generator().manual_seed(h * 7 + w)
    x = _r(torch.randn(n, 128, h, w, generator=g))
    M = n * h * w
    wq, wk, wv, ws = (_r(torch.randn(c, 128, generator=g) / 11) for c in (ck, ck, cv, ns))
    stack = torch.zeros(512, 128)
    stack[:ck], stack[ckp:ckp + ck], stack[ckp + ck:ckp + ck + cv], stack[ckp + ck + cv:ckp + ck + cv + ns] = wq, wk, wv, ws
    wf = torch.cat([_head2_frag(stack[256 * i:256 * (i + 1)], BF16) for i in range(2)], 0).contiguous().to(dev)
    xin = _nhwc16(x, 136)
    q = torch.full((M, ckp + 8), 512.0, device=dev, dtype=BF16)
    kv = torch.full((M, ck + cv + 8), 512.0, device=dev, dtype=BF16)
    sg = torch.full((M, 8), 512.0, device=dev)
    d = _hip.QkvsBf16Desc()
    d.inp, d.in_cs, d.M, d.wf = xin.data_ptr(), 136, M, wf.data_ptr()
    d.q, d.q_cs, d.q_rows = q.data_ptr(), ckp + 8, ckp
    d.kv, d.kv_cs, d.kv_rows = kv.data_ptr(), ck + cv + 8, ck + cv
    d.s, d.s_cs, d.s_rows = sg.data_ptr(), 8, ns
    _hip.check(L.m3d_anab_qkvs_bf16_forward(ctypes.byref(d), _st()))
    torch.cuda.synchronize()
    xf = x.permute(0, 2, 3, 1).reshape(M, 128)
    for name, got, ref in (("q", q[:, :ck], xf @ wq.T), ("kv", kv[:, :ck + cv], xf @ torch.cat([wk, wv]).T)):
        e = (got.float().cpu() - ref).abs()
        assert (e <= 2.0 ** -8 * ref.abs() + 1e-4).all(), (name, e.max().item())
    assert (q[:, ck:ckp].float() == 0).all() and (q[:, ckp:].float() == 512.0).all() and (kv[:, ck + cv:].float() == 512.0).all()
    es = (sg[:, :ns].cpu() - torch.sigmoid(xf @ ws.T)).abs().max().item()
    assert es < 2e-4 and (sg[:, ns:] == 512.0).all(), es


@pytest.mark.parametrize("n,h,w,cout", [(2, 8, 16, 144), (1, 13, 21, 144), (3, 48, 160, 144), (1, 8, 8, 256), (2, 4, 16, 20)])
def test_head_tail2_bf16_matches_torch_chain(n, h, w, cout):
    """m3d_head_tail2_bf16_forward (cls.3 + cls.6: 256 -> 256 + affine + LeakyReLU -> 256 -> Cout + affine in one launch) against the
    torch chain on the bf16-rounded input / weights (hidden map rounded to bf16 in the reference; the kernel keeps it in fp16 and folds
    the scales into the weights: inside the bound of the 3-layer heads).  Ragged tiles, several tiles per workgroup, Cout = 256 / 20."""
    from m3dssd_amd import _hip
    from m3dssd_amd.engine_bf16 import pack_tail2
    L, dev = _hip.lib(), _dev()
    g = torch.Generator().manual_seed(n * 100 + h + cout)
    x = _r(torch.randn(n, 256, h, w, generator=g))
    M, HW = n * h * w, h * w
    xin = _nhwc16(x, 264)
    wa = _r(torch.randn(256, 256, generator=g) / 16)
    wb = _r(torch.randn(cout, 256, generator=g) / 16)
    sa, ta = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    sb, tb = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    out = torch.full((n, cout + 1, HW), 512.0, device=dev)
    pk = pack_tail2(wa, sa, ta, wb, sb, tb, dev)
    d = _hip.Tail2Bf16Desc()
    d.inp, d.in_cs, d.M = xin.data_ptr(), 264, M
    d.waf, d.wbf, d.t1, d.t2 = (t.data_ptr() for t in pk)
    d.Cout, d.out, d.out_img_stride, d.HW = cout, out.data_ptr(), (cout + 1) * HW, HW
    _hip.check(L.m3d_head_tail2_bf16_forward(ctypes.byref(d), _st()))
    torch.cuda.synchronize()
    got = out.cpu()
    assert (got[:, cout] == 512.0).all()
    xf = x.permute(0, 2, 3, 1).reshape(M, 256)
    h1 = _r(F.leaky_relu(xf @ wa.T * sa + ta, 0.01))
    ref = (h1 @ wb.T * sb + tb).view(n, HW, cout).permute(0, 2, 1)
    e = (got[:, :cout] - ref).abs().max().item()
    assert e < 4e-3 * (1.0 + ref.abs().max().item()), e


@pytest.mark.parametrize("B,h,w,keys", [(2, 8, 16, 337), (1, 16, 40, 337), (3, 8, 32, 85)])
def test_anab_attend_bf16_matches_torch(B, h, w, keys):
    """m3d_anab_attend_bf16 (logits + softmax + P.V + residual + affine + LeakyReLU in one launch, attention.py:207-211) against
    torch fp32 on the bf16-rounded operands.  The kernel rounds exp(S - max) to bf16 and divides by the row sum in fp32."""
    from m3dssd_amd import _hip
    L, dev = _hip.lib(), _dev()
    g = torch.Generator().manual_seed(B * 100 + keys)
    HW, ck, ckp, cv = h * w, 168, 192, 128
    kp = (keys + 63) // 64 * 64
    q = torch.zeros(B * HW, ckp)
    q[:, :ck] = _r(torch.randn(B * HW, ck, generator=g) * 0.5)
    khat = torch.zeros(B, kp, ckp)
    khat[:, :keys, :ck] = _r(torch.randn(B, keys, ck, generator=g) * 0.3)
    vhat = torch.zeros(B, cv, kp)
    vhat[:, :, :keys] = _r(torch.randn(B, cv, keys, generator=g))
    vhat[:, :, keys:] = 7.0                                  # rows past `keys` must be ignored
    res = _r(torch.randn(B * HW, cv, generator=g))
    scale, shift = torch.rand(cv, generator=g) + 0.5, torch.randn(cv, generator=g) * 0.1
    S = torch.einsum("bpc,bkc->bpk", q.view(B, HW, ckp), khat)[:, :, :keys]
    Pm = torch.softmax(S, dim=-1)
    ref = torch.einsum("bpk,bck->bpc", Pm, vhat[:, :, :keys]).reshape(B * HW, cv)
    ref = F.leaky_relu((ref + res) * scale + shift, 0.01)
    dq, dk, dv, dr = (t.to(BF16).contiguous().to(dev) for t in (q, khat, vhat, res))
    dsc, dsh = scale.to(dev), shift.to(dev)
    out = torch.full((B * HW, cv + 8), 512.0, device=dev, dtype=BF16)
    _hip.check(L.m3d_anab_attend_bf16(dq.data_ptr(), ckp, dk.data_ptr(), dv.data_ptr(), B, HW, ckp, keys, kp, cv, dr.data_ptr(), cv,
                                      dsc.data_ptr(), dsh.data_ptr(), 1, out.data_ptr(), cv + 8, _st()))
    torch.cuda.synchronize()
    assert (out[:, cv:].float() == 512.0).all()
    got = out[:, :cv].float().cpu()
    sc = ref.abs().max().item()
    bad = (got - ref).abs() > 2.0 ** -7 * ref.abs() + 2e-3 * sc
    assert not bad.any(), ((got - ref).abs().max().item(), sc, int(bad.sum()))
    # argument validation
    assert L.m3d_anab_attend_bf16(dq.data_ptr(), ckp, dk.data_ptr(), dv.data_ptr(), B, HW + 1, ckp, keys, kp, cv, None, 0, None, None, 0,
                                  out.data_ptr(), cv + 8, _st()) == -1


@pytest.mark.parametrize("shape", [(2, 128, 16, 40, 128, 3, 1), (1, 256, 8, 20, 128, 3, 1), (1, 512, 6, 10, 256, 3, 1),
                                   (2, 128, 16, 40, 128, 1, 0), (1, 64, 9, 13, 64, 3, 1)])
def test_dcn_bf16_matches_oracle(shape):
    """Deformable mode against oracle/dcn.py on the bf16-rounded input / weights; offsets and masks are fp32 in both.  The
    kernel rounds the modulated bilinear sample to bf16 before the GEMM (one extra 2^-9 relative rounding per sample), so the
    bound is sqrt(K)-scaled bf16 noise: 1% of the output scale."""
    from oracle import dcn as odcn
    n, c, h, w, co, k, pad = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = _r(torch.randn(n, c, h, w, generator=g))
    wt = _r(torch.randn(co, c, k, k, generator=g) / (c * k * k) ** 0.5)
    b = torch.randn(co, generator=g) * 0.1
    off = torch.randn(n, 2 * k * k, h, w, generator=g) * 3.0
    off[0, 0, 0, 0] = -1.0 + pad
    m = torch.rand(n, k * k, h, w, generator=g)
    ref = odcn.dcn_v2_forward(x, off, m, wt, b, 1, pad, 1, 1)
    om = torch.cat([off, m], 1).permute(0, 2, 3, 1).contiguous()            # NHWC [.., 3*k*k]
    om = torch.cat([om, torch.zeros(n, h, w, (-om.shape[-1]) % 4)], -1).contiguous()
    got = _run_conv(x, wt, b, None, 1, pad, 0, None, 0, -1, 0, om)
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    _log("dcn_bf16", dict(shape=list(shape), err=err, scale=scale))
    assert err < 1e-2 * scale + 2.0 ** -8 * scale
    # fp32 output mode isolates the sample rounding from the output rounding
    got32 = _run_conv(x, wt, b, None, 1, pad, 0, None, 0, -1, 1, om)
    assert (got32 - ref).abs().max().item() < 1e-2 * scale


@pytest.mark.parametrize("n,h,w,act,with_res", [(2, 16, 40, 0, True), (1, 9, 13, 1, False), (2, 48, 160, 0, True), (3, 8, 16, 1, True)])
def test_dcn1x1_bf16_kernel_matches_oracle_and_the_generic_tile(n, h, w, act, with_res):
    """csrc/bf16_dcn1x1.hip (center_align's op: 1x1 DCNv2 128 -> 128 + bias (+ input as residual), feturealign_mgpu.py:48-99) against
    oracle/dcn.py on the bf16-rounded operands, and against the generic deformable tile (the fp32 output mode of the same
    descriptor runs there; rounded to bf16 on the host): same corner rules (dcn_corners), same fp32 combine, same K order -- bit-identical
    except for a handful of entries (< 1e-4 of them, <= 0.4 % of the output scale).
    Several tiles, a ragged single tile (117 pixels), the full-size map; offsets on the image border, far outside, NaN and inf."""
    from oracle import dcn as odcn
    g = torch.Generator().manual_seed(n * 100 + h)
    c = co = 128
    x = _r(torch.randn(n, c, h, w, generator=g) + 0.5)
    wt = _r(torch.randn(co, c, 1, 1, generator=g) / c ** 0.5)
    b = torch.randn(co, generator=g) * 0.1
    off = torch.randn(n, 2, h, w, generator=g) * 2.5
    off[0, :, 0, 0] = torch.tensor([-1.0, -1.0])               # exactly on the "contributes nothing" border
    off[0, :, 0, 1] = torch.tensor([-0.5, float(w)])           # column far outside
    off[0, :, 1, 0] = torch.tensor([float(h) - 1.0, 0.25])     # row H: outside
    off[0, :, 1, 1] = torch.tensor([1e9, 0.0])
    m = torch.rand(n, 1, h, w, generator=g)
    ref = odcn.dcn_v2_forward(x, off, m, wt, b, 1, 0, 1, 1)
    if with_res:
        ref = ref + x
    if act:
        ref = F.leaky_relu(ref, 0.01)
    off[0, :, 2, 2] = float("nan")                             # non-finite offsets: the sample contributes nothing (dcn_v2_im2col_cuda.cu:165)
    off[0, 0, 2, 3] = float("inf")
    for yy, xx in ((2, 2), (2, 3)):
        r = b.view(-1) + (x[0, :, yy, xx] if with_res else 0.0)
        ref[0, :, yy, xx] = F.leaky_relu(r, 0.01) if act else r
    om = torch.cat([off, m, torch.zeros(n, 1, h, w)], 1).permute(0, 2, 3, 1).contiguous()
    res = x if with_res else None
    got = _run_conv(x, wt, b, None, 1, 0, act, res, 0, -1, 0, om, variant=6)
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    _log("dcn1x1_bf16", dict(shape=[n, h, w], err=err, scale=scale))
    assert torch.isfinite(got).all() and err < 1e-2 * scale + 2.0 ** -8 * scale
    gen32 = _run_conv(x, wt, b, None, 1, 0, act, res, 0, -1, 1, om, variant=0)          # fp32 NHWC output: the generic tile
    gen = gen32.to(BF16).float()
    diff = (got - gen).abs()
    # measured on the full-size map: 12 of 1.97 M entries differ, by up to 2^-7 (a sample that rounds to the neighbouring bf16 in one
    # of the two kernels: their fp32 blends contract differently); every other entry is bit-identical
    assert float(diff.max()) <= 4e-3 * scale and float((diff > 0).float().mean()) < 1e-4, (diff.max().item(), float((diff > 0).float().mean()))
    again = _run_conv(x, wt, b, None, 1, 0, act, res, 0, -1, 0, om, variant=6)
    assert torch.equal(got, again)


def _dcn_case(shape, off_std, seed=0, clamp=None):
    n, c, h, w, co = shape
    g = torch.Generator().manual_seed(sum(shape) + seed)
    x = _r(torch.randn(n, c, h, w, generator=g) + 0.5)
    wt = _r(torch.randn(co, c, 3, 3, generator=g) / (c * 9) ** 0.5)
    b = torch.randn(co, generator=g) * 0.1
    off = torch.randn(n, 18, h, w, generator=g) * off_std
    if clamp is not None:
        off = off.clamp(-clamp, clamp)
    m = torch.rand(n, 9, h, w, generator=g)
    om = torch.cat([off, m], 1).permute(0, 2, 3, 1).contiguous()
    om = torch.cat([om, torch.zeros(n, h, w, 5)], -1).contiguous()          # pixel stride 32 like the engine's maps
    return x, wt, b, off, m, om


@pytest.mark.parametrize("shape,variant,off_std,clamp", [
    ((2, 128, 16, 32, 128), 4, 1.5, None),        # 16 x 16 patches, offsets up to ~6
    ((1, 128, 48, 160, 128), 4, 2.0, 8.9),        # the full-size backbone map, window radius 9 (the largest)
    ((2, 256, 24, 80, 256), 3, 1.2, 5.9),         # 8 x 16 patches, two channel tiles, radius 6
    ((3, 64, 8, 16, 128), 3, 0.3, None),          # one patch per image: every window crosses all four image borders
    ((1, 32, 16, 16, 100), 4, 1.0, None),         # ragged Cout (pad 128), one chunk
])
def test_dcn_bf16_patch_kernel_matches_oracle(shape, variant, off_std, clamp):
    """The LDS-patch DCNv2 kernel (csrc/bf16_dcn_patch.hip: fp16 window in LDS, v_pk_fma_f16 bilinear combine straight into the
    B operand of v_mfma_f32_32x32x16_f16) against oracle/dcn.py on the bf16-rounded operands, and against the implicit-GEMM
    kernel it replaces.  Its intermediate (the fp16 sample, 11 significant bits) is finer than the implicit-GEMM kernel's (bf16
    sample, 8 bits): the same 1 % of the output scale bounds both; the two kernels agree to that bound as well."""
    from oracle import dcn as odcn
    x, wt, b, off, m, om = _dcn_case(shape, off_std, 1, clamp)
    ref = odcn.dcn_v2_forward(x, off, m, wt, b, 1, 1, 1, 1)
    scale = ref.abs().max().item()
    got32 = _run_conv(x, wt, b, None, 1, 1, 0, None, 0, -1, 1, om, variant=variant, patch=True)
    err = (got32 - ref).abs().max().item()
    old32 = _run_conv(x, wt, b, None, 1, 1, 0, None, 0, -1, 1, om, variant=0)
    err_old = (old32 - ref).abs().max().item()
    _log("dcn_bf16_patch", dict(shape=list(shape), err=err, err_implicit_gemm=err_old, scale=scale))
    assert err < 1e-2 * scale
    assert err <= max(err_old * 1.25, 2e-3 * scale)            # not less accurate than the kernel it replaces
    # bf16 NHWC output with BatchNorm + LeakyReLU + residual through the shared epilogue
    g = torch.Generator().manual_seed(3)
    co = shape[4]
    bn = (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1,
          torch.rand(co, generator=g) + 0.5)
    res = _r(torch.randn(shape[0], co, shape[2], shape[3], generator=g))
    ref2 = F.leaky_relu(F.batch_norm(odcn.dcn_v2_forward(x, off, m, wt, b, 1, 1, 1, 1), bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
                        + res, 0.01)
    got2 = _run_conv(x, wt, b, bn, 1, 1, 1, res, 0, -1, 0, om, variant=variant, patch=True)
    s2 = ref2.abs().max().item()
    assert (got2 - ref2).abs().max().item() < 1e-2 * s2 + 2.0 ** -8 * s2


def test_dcn_bf16_patch_kernel_hands_over_when_the_window_does_not_fit():
    """The decision is taken on the device PER PIXEL TILE from the tile's largest |offset|: inside the radius the patch kernel does the
    tile, one offset beyond it (or a NaN) and the tile raises its flag -- the implicit-GEMM kernel behind it then recomputes the
    128-pixel tiles that touch it, bit-identical to a launch without the patch kernel, and leaves the rest of the map alone."""
    from oracle import dcn as odcn
    shape = (2, 64, 16, 32, 128)
    x, wt, b, off, m, om = _dcn_case(shape, 1.0, 2, 4.0)
    base = _run_conv(x, wt, b, None, 1, 1, 0, None, 0, -1, 0, om, variant=0)
    fit = _run_conv(x, wt, b, None, 1, 1, 0, None, 0, -1, 0, om, variant=4, patch=True)
    assert not torch.equal(fit, base)                           # a different kernel did the work (fp16 sample vs bf16 sample)
    for bad in (10.25, -37.0, float("nan")):
        om2 = om.clone()
        om2[1, 7, 9, 5] = bad                                   # dw of tap 2 at one pixel of image 1, patch tile (rows 0-15, columns 0-15)
        base2 = _run_conv(x, wt, b, None, 1, 1, 0, None, 0, -1, 0, om2, variant=0)
        got2 = _run_conv(x, wt, b, None, 1, 1, 0, None, 0, -1, 0, om2, variant=4, patch=True)
        # image 1: every 128-pixel tile (4 rows of 32) touches the flagged patch tile -> the implicit-GEMM kernel's bits;
        # image 0: untouched -> the patch kernel's bits
        assert torch.equal(got2[1], base2[1]), bad
        assert torch.equal(got2[0], fit[0]) and not torch.equal(got2[0], base2[0]), bad
    # a flagged tile on a map with several 128-pixel tiles per patch-tile row: only the touching ones are recomputed, every pixel
    # carries one of the two kernels' bits, the flagged tile itself the implicit-GEMM kernel's
    shape = (1, 64, 32, 160, 128)
    x, wt, b, off, m, om = _dcn_case(shape, 1.0, 3, 4.0)
    om2 = om.clone()
    om2[0, 20, 100, 3] = 15.5                                   # patch tile rows 16-31, columns 96-111
    base2 = _run_conv(x, wt, b, None, 1, 1, 0, None, 0, -1, 0, om2, variant=0)
    fit = _run_conv(x, wt, b, None, 1, 1, 0, None, 0, -1, 0, om, variant=4, patch=True)
    got2 = _run_conv(x, wt, b, None, 1, 1, 0, None, 0, -1, 0, om2, variant=4, patch=True)
    assert torch.equal(got2[0, :, 16:32, 96:112], base2[0, :, 16:32, 96:112])
    assert torch.equal(got2[0, :, 0:16], fit[0, :, 0:16])       # rows 0-15: no 128-pixel tile of theirs reaches row 16
    same_base, same_fit = (got2 == base2).all(1), (got2 == fit).all(1)
    assert bool((same_base | same_fit).all()) and not bool(same_base.all())
    # the recomputation rewrites whole 128-pixel tiles: an in-place residual (res == out) would be added twice -> refused up front
    rc, msg = _run_conv(x, wt, b, None, 1, 1, 0, None, 0, -1, 0, om2, patch=True, alias_res_out=True)
    assert rc != 0 and "res != out" in msg
    # exactly at the radius it still fits
    om3 = om.clone()
    om3[0, 3, 4, 0] = 9.0
    off3 = om3[..., :18].permute(0, 3, 1, 2).contiguous()
    ref3 = odcn.dcn_v2_forward(x, off3, m, wt, b, 1, 1, 1, 1)
    got3 = _run_conv(x, wt, b, None, 1, 1, 0, None, 0, -1, 1, om3, variant=4, patch=True)
    base3 = _run_conv(x, wt, b, None, 1, 1, 0, None, 0, -1, 1, om3, variant=0)
    assert not torch.equal(got3, base3)
    assert (got3 - ref3).abs().max().item() < 1e-2 * ref3.abs().max().item()


def test_dcn_bf16_patch_kernel_border_positions():
    """Exact border positions through the zero-padded window: a 3x3 deformable conv whose centre tap samples the grid of
    _border_grid_offsets (exactly -1, just inside, between rows, the last row, past it, exactly H) while the other taps have
    mask 0 -- the patch kernel must reproduce the reference's corner rules without any per-corner test."""
    from oracle import dcn as odcn
    h, w, c, co = 16, 32, 32, 128
    g = torch.Generator().manual_seed(9)
    x = _r(torch.randn(1, c, h, w, generator=g) + 3.0)
    wt = _r(torch.randn(co, c, 3, 3, generator=g) / (9 * c) ** 0.5)
    b = torch.zeros(co)
    off = torch.zeros(1, 18, h, w)
    m = torch.zeros(1, 9, h, w)
    o1 = _border_grid_offsets(h, w)                              # absolute position hs[i % 8] for a 1x1 tap at (i, j)
    off[:, 8:10] = o1                                            # centre tap (index 4) of a 3x3 / pad 1 kernel sits at (i, j) too
    m[:, 4] = 1.0
    # keep the launch inside the patch radius: the grid asks for |offset| up to 16 on this map; drop the taps that ask for more than the radius 9
    keep = (off.abs() <= 8.9).all(1, keepdim=True)
    off = off * keep
    m = m * keep
    ref = odcn.dcn_v2_forward(x, off, m, wt, b, 1, 1, 1, 1)
    om = torch.cat([off, m, torch.zeros(1, 5, h, w)], 1).permute(0, 2, 3, 1).contiguous()
    got = _run_conv(x, wt, b, None, 1, 1, 0, None, 0, -1, 1, om, variant=4, patch=True)
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 1e-2 * scale
    zero_rows = ref.abs().amax(dim=(0, 1, 3)) == 0
    assert zero_rows.any() and torch.equal(got[0, :, zero_rows, :], torch.zeros_like(got[0, :, zero_rows, :]))


def _border_grid_offsets(h, w):
    """1x1 deformable conv whose output pixel (i, j) samples the input at (hs[i % 8], ws[j % 8]): every combination of the
    in / out decisions of dcn_v2_im2col_cuda.cu:129-178 -- exactly -1 (out, the gate is strict), just inside, between rows, 0,
    the last row, past the last row (one corner row dropped), just below H, exactly H (out)."""
    hs = [-1.0, -0.999, -0.5, 0.0, h - 1.0, h - 0.5, h - 0.001, float(h)]
    ws = [-1.0, -0.999, -0.5, 0.0, w - 1.0, w - 0.5, w - 0.001, float(w)]
    off = torch.zeros(1, 2, h, w)
    for i in range(h):
        for j in range(w):
            off[0, 0, i, j] = hs[i % 8] - i
            off[0, 1, i, j] = ws[j % 8] - j
    return off


def test_dcn_bf16_sampling_border_grid():
    """The corner in / out decisions of the sampling code (csrc/common.h dcn_corners: sign-smear masks instead of compares) on a
    grid of exact border positions, bf16 kernel vs oracle/dcn.py, uniform-K and general-K paths (Cin 64 / 24)."""
    from oracle import dcn as odcn
    for c in (64, 24):
        h, w, co = 16, 24, 32
        g = torch.Generator().manual_seed(5 + c)
        x = _r(torch.randn(1, c, h, w, generator=g) + 3.0)                  # offset from zero: a dropped corner shows
        wt = _r(torch.randn(co, c, 1, 1, generator=g) / c ** 0.5)
        b = torch.zeros(co)
        off = _border_grid_offsets(h, w)
        m = torch.ones(1, 1, h, w)
        ref = odcn.dcn_v2_forward(x, off, m, wt, b, 1, 0, 1, 1)
        om = torch.cat([off, m, torch.zeros(1, 1, h, w)], 1).permute(0, 2, 3, 1).contiguous()
        got32 = _run_conv(x, wt, b, None, 1, 0, 0, None, 0, -1, 1, om)
        scale = ref.abs().max().item()
        assert (got32 - ref).abs().max().item() < 1e-2 * scale, c
        # rows / columns sampled exactly at -1 or at H / W contribute nothing at all
        assert torch.equal(got32[0, :, 0::8, :], torch.zeros_like(got32[0, :, 0::8, :])) and \
            torch.equal(got32[0, :, 7::8, :], torch.zeros_like(got32[0, :, 7::8, :])) and \
            torch.equal(got32[0, :, :, 0::8], torch.zeros_like(got32[0, :, :, 0::8])) and \
            torch.equal(got32[0, :, :, 7::8], torch.zeros_like(got32[0, :, :, 7::8]))


def test_bf16_helpers_match_torch():
    from m3dssd_amd import _hip
    L = _hip.lib()
    dev = _dev()
    g = torch.Generator().manual_seed(8)
    # maxpool (exact) on a channel-sliced view
    x = _r(torch.randn(2, 32, 8, 12, generator=g))
    xin = _nhwc16(x, 48)
    out = torch.zeros(2, 4, 6, 40, device=dev, dtype=BF16)
    _hip.check(L.m3d_maxpool2x2_bf16(xin.data_ptr(), 48, out.data_ptr(), 40, 2, 8, 12, 32, _st()))
    assert torch.equal(out[..., :32].float().permute(0, 3, 1, 2).cpu(), F.max_pool2d(x, 2, 2))
    assert (out[..., 32:] == 0).all()
    # depthwise ConvTranspose2d(4, 2, 1) + skip: the grid-stride form (16 channels) and the row form (64 / 128 / 256 channels, ragged widths)
    for c, h, w in ((16, 5, 7), (128, 5, 7), (256, 3, 9), (64, 6, 5)):
        x = _r(torch.randn(2, c, h, w, generator=g))
        wt = torch.rand(c, 1, 4, 4, generator=g)
        skip = _r(torch.randn(2, c, 2 * h, 2 * w, generator=g))
        ref = F.conv_transpose2d(x, wt, None, stride=2, padding=1, groups=c) + skip
        xin, sk = _nhwc16(x), _nhwc16(skip)
        wd = wt[:, 0].permute(1, 2, 0).contiguous().to(dev)
        out = torch.zeros(2, 2 * h, 2 * w, c, device=dev, dtype=BF16)
        _hip.check(L.m3d_upsample2x_add_bf16(xin.data_ptr(), c, wd.data_ptr(), sk.data_ptr(), c, out.data_ptr(), c, 2, h, w, c, _st()))
        got = out.float().permute(0, 3, 1, 2).cpu()
        assert ((got - ref).abs() <= 2.0 ** -8 * ref.abs() + 1e-6).all(), (c, h, w)
    # fp32 -> bf16 (round to nearest even, like torch)
    v = torch.randn(4096, generator=g) * 100
    o = torch.zeros(4096, device=dev, dtype=BF16)
    vd = v.to(dev)
    _hip.check(L.m3d_f32_to_bf16(vd.data_ptr(), o.data_ptr(), 4096, _st()))
    assert torch.equal(o.cpu(), v.to(BF16))
    # softmax rows -> bf16
    lg = torch.randn(50, 384, generator=g) * 3
    p = torch.full((50, 384), 9.0, device=dev, dtype=BF16)
    ld = lg.to(dev)
    _hip.check(L.m3d_softmax_rows_bf16(ld.data_ptr(), 50, 337, 384, p.data_ptr(), 384, _st()))
    ref = torch.softmax(lg[:, :337], -1)
    got = p.float().cpu()
    assert ((got[:, :337] - ref).abs() <= 2.0 ** -8 * ref + 1e-7).all() and (got[:, 337:] == 0).all()
    # stem: fp32 image and uint8 frames (Preprocess fused) against the fp32 stem on the same input
    w7 = torch.randn(16, 3, 7, 7, generator=g) / 12
    sc, sh = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.1
    img = torch.randn(2, 3, 16, 64, generator=g)
    ref = F.leaky_relu(F.conv2d(img, w7, None, padding=3) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), 0.01)
    wd = w7.permute(2, 3, 1, 0).contiguous().to(dev)
    out = torch.zeros(2, 16, 64, 16, device=dev, dtype=BF16)
    imd, scd, shd = img.to(dev), sc.to(dev), sh.to(dev)
    _hip.check(L.m3d_stem_conv7x7_bf16(imd.data_ptr(), 0, 0, 0, None, None, wd.data_ptr(), scd.data_ptr(), shd.data_ptr(),
                                       out.data_ptr(), 16, 2, 16, 64, _st()))
    got = out.float().permute(0, 3, 1, 2).cpu()
    assert ((got - ref).abs() <= 2.0 ** -8 * ref.abs() + 1e-4).all()
    from oracle import preprocess as opre
    frames = torch.randint(0, 256, (2, 12, 50, 3), generator=g, dtype=torch.uint8)
    conf = synth.synth_conf((16, 64), 0, batch_size=2, device="cpu")
    mean_np, stds_np = np.asarray(conf.image_means, dtype=np.float32), np.asarray(conf.image_stds, dtype=np.float32)
    pre = torch.from_numpy(np.stack([opre.preprocess(f.numpy(), (16, 64), mean_np, stds_np) for f in frames]))
    ref = F.leaky_relu(F.conv2d(pre, w7, None, padding=3) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), 0.01)
    mean3 = (ctypes.c_float * 3)(*[float(v) for v in conf.image_means])
    stds3 = (ctypes.c_float * 3)(*[float(v) for v in conf.image_stds])
    fd = frames.to(dev)
    _hip.check(L.m3d_stem_conv7x7_bf16(fd.data_ptr(), 1, 12, 50, mean3, stds3, wd.data_ptr(), scd.data_ptr(), shd.data_ptr(),
                                       out.data_ptr(), 16, 2, 16, 64, _st()))
    got = out.float().permute(0, 3, 1, 2).cpu()
    assert ((got - ref).abs() <= 2.0 ** -8 * ref.abs() + 1e-4).all()


@pytest.mark.parametrize("n,H,W,u8", [(2, 32, 128, False), (1, 48, 96, False), (2, 32, 128, True), (2, 96, 192, False),
                                      (1, 64, 80, False), (3, 96, 256, True)])
def test_fused_frontend_bf16_matches_torch_chain(n, H, W, u8):
    """m3d_frontend2_bf16_forward (stem -> level0 -> level1 in one launch, intermediates in LDS) against the torch chain with
    bf16-rounded weights and the two intermediates rounded to bf16 (the kernel keeps them in fp16 and folds the BatchNorm scales
    into fp16 weights: inside the same bound); image borders, several tiles per image incl. interior ones (96 x 192 / 96 x 256:
    the mask-free code path), a width that is not a multiple of the tile, and the uint8 input path."""
    from m3dssd_amd import _hip
    from m3dssd_amd.engine_bf16 import pack_frontend_f16
    L = _hip.lib()
    dev = _dev()
    g = torch.Generator().manual_seed(H + W)
    ws = _r(torch.randn(16, 3, 7, 7, generator=g) / 12)
    w0 = _r(torch.randn(16, 16, 3, 3, generator=g) / 12)
    w1 = _r(torch.randn(32, 16, 3, 3, generator=g) / 12)
    aff = [(torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1) for c in (16, 16, 32)]
    conf = synth.synth_conf((H, W), 0, batch_size=n, device="cpu")
    if u8:
        from oracle import preprocess as opre
        frames = torch.randint(0, 256, (n, H - 5, W - 9, 3), generator=g, dtype=torch.uint8)
        mean_np, stds_np = np.asarray(conf.image_means, dtype=np.float32), np.asarray(conf.image_stds, dtype=np.float32)
        img = torch.from_numpy(np.stack([opre.preprocess(f.numpy(), (H, W), mean_np, stds_np) for f in frames]))
        src = frames.to(dev)
    else:
        img = torch.randn(n, 3, H, W, generator=g)
        src = img.to(dev)

    def stage(x, w, sc, sh, stride, pad):
        return _r(F.leaky_relu(F.conv2d(x, w, None, stride=stride, padding=pad) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), 0.01))
    ref = stage(stage(stage(_r(img), ws, *aff[0], 1, 3), w0, *aff[1], 1, 1), w1, *aff[2], 2, 1)
    out = torch.full((n, H // 2, W // 2, 40), 512.0, device=dev, dtype=BF16)
    mean3 = (ctypes.c_float * 3)(*[float(v) for v in conf.image_means])
    stds3 = (ctypes.c_float * 3)(*[float(v) for v in conf.image_stds])
    f2 = pack_frontend_f16(ws, aff[0], w0, aff[1], w1, aff[2], dev)
    _hip.check(L.m3d_frontend2_bf16_forward(src.data_ptr(), 1 if u8 else 0, H - 5 if u8 else 0, W - 9 if u8 else 0, mean3,
                                            stds3, f2[0].data_ptr(), f2[1].data_ptr(), f2[2].data_ptr(), f2[3].data_ptr(),
                                            f2[4].data_ptr(), f2[5].data_ptr(), out.data_ptr(), 40, n, H, W, _st()))
    torch.cuda.synchronize()
    assert (out[..., 32:].float() == 512.0).all()
    got = out[..., :32].float().permute(0, 3, 1, 2).cpu()
    err = (got - ref).abs()
    tol = 2.0 ** -7 * ref.abs() + 4e-3 * ref.abs().max()      # an intermediate may round to the neighbouring bf16
    assert (err <= tol).all(), (err.max().item(), ref.abs().max().item(), int((err > tol).sum()))


# ------------------------------------------------------------------------------------ whole network
def _net(crop, B, dtype):
    from model.M3d_inference_align import build
    conf = synth.synth_conf(crop, 0, batch_size=B, device="cuda:0")
    net = build(conf, "test")
    net.load_state_dict(synth.synth_state_dict(0), strict=True)
    return net.to(_dev()).set_compute_dtype(dtype), conf


def _bf16_vs_oracle(net, plan, x, outs, rows, crop):
    """Engine outputs `outs` = (cls, prob, bbox_2d, bbox_3d) of the frames `rows` of the plan's batch (CPU tensors) against the fp32
    CPU oracle on those frames `x`, the engine's top-1 anchor / hard-mask decisions injected -> report dict."""
    from oracle import model_cpu
    sd = synth.synth_state_dict(0)
    n = len(rows)
    fh, fw = crop[0] // 8, crop[1] // 8
    ind = plan.named["sel_idx"].view(-1, 1, fh, fw)[rows].long().cpu()
    prob_sel = plan.named["sel_prob"].view(-1, 1, fh, fw)[rows].cpu()
    cconf = synth.synth_conf(crop, 0, batch_size=n, device="cpu")
    taps, taps_free = {}, {}
    with torch.no_grad():
        free = model_cpu.rpn_forward(sd, cconf, x, taps_free)
        inj = model_cpu.rpn_forward(sd, cconf, x, taps, inject={"sel": {"ind": ind, "hard": (prob_sel > 0.5).float()}})
    cls, prob, b2, b3 = outs
    fg = taps_free["fg_prob"]
    o_mask, o_ind = fg.max(dim=1, keepdim=True)
    diff = o_ind != ind
    rep = {"crop": list(crop), "frames": n, "n_idx": int(diff.sum()), "n_flip": int(((o_mask > 0.5) != (prob_sel > 0.5)).sum()),
           "pixels": int(ind.numel())}
    rep["flip_rate"] = rep["n_idx"] / rep["pixels"]
    # the oracle's own margin at the pixels where the engine chose another anchor (all inside the probability tolerance)
    rep["flip_margin_max"] = (o_mask - torch.gather(fg, 1, ind))[diff].abs().max().item() if diff.any() else 0.0
    for name in ("level2", "level5", "feats0", "feats", "feats_align2d", "feats_align3d", "feats_gl"):
        got = plan.named[name].torch_nchw()[rows].cpu()
        ref = (taps_free if name in ("level2", "level5", "feats0") else taps)[name]
        rep["rel_" + name] = ((got - ref).abs().max() / ref.abs().max()).item()
    rep["cls"] = (cls - free[0]).abs().max().item()
    rep["prob"] = (prob - inj[1]).abs().max().item()
    for nm, got, ref in (("bbox_2d", b2, inj[2]), ("bbox_3d", b3, inj[3])):
        e = (got - ref).abs().view(-1, got.shape[-1])
        rep[nm] = e.max().item()
        rep[nm + "_cols"] = {c: [e[:, j].max().item(), torch.quantile(e[:, j][::3].float(), 0.999).item(),
                                 e[:, j].pow(2).mean().sqrt().item()] for j, c in enumerate(BF16_COLS[nm])}
    assert torch.isfinite(b3).all() and torch.isfinite(cls).all()
    return rep


def _assert_bf16_report(rep, full_size):
    """The per-column bounds (1.3 x measured) hold at the size they were measured at; smaller maps are looser by construction (fewer
    rows for the maxima) and are held to the same table."""
    assert rep["prob"] < BF16_PROB_TOL and rep["prob"] <= BF16_GUARD * BF16_PROB_MEASURED + (0.0 if full_size else 0.01), rep
    for nm, cols in BF16_COLS.items():
        for c in cols:
            mx, p999, rms = rep[nm + "_cols"][c]
            tmx, tp, tr = (BF16_GUARD * v for v in BF16_MEASURED[nm][c])
            assert mx <= tmx and p999 <= tp and rms <= tr, (nm, c, (mx, p999, rms), (tmx, tp, tr))
    # decisions differ from the free-running fp32 oracle's only where its own margin is within bf16 noise, and not more often
    # than measured (x 1.3; the small map has 1280 pixels: its rate is noisier and gets the absolute slack of 10 pixels)
    assert rep["flip_margin_max"] < BF16_PROB_TOL, rep
    assert rep["n_idx"] <= BF16_GUARD * BF16_FLIP_RATE_MEASURED * rep["pixels"] + (0 if full_size else 10), rep


@pytest.mark.parametrize("crop,B,pad", [((128, 320), 2, False), ((384, 1280), 2, True)])
def test_bf16_network_matches_fp32_oracle_within_stated_tolerance(crop, B, pad):
    """bf16 engine vs the fp32 CPU oracle, the engine's discrete decisions (top-1 anchor, hard mask) injected into the
    oracle.  Logs max / p99.9 / rms of every output column, the stage taps and the top-1-anchor flip rate; asserts the per-column
    bounds (1.3 x measured)."""
    net, conf = _net(crop, B, "bf16")
    x = synth.synth_frames(B, crop, 1234, pad_right_third=pad)
    with torch.no_grad():
        outs = [t.cpu() for t in net(x.to(_dev()))[:4]]
    eng = net.engine()
    assert type(eng).__name__ == "EngineBF16"
    plan = eng.plan_for(B, *crop)
    assert all(not k[1].startswith(("igemm", "wino", "conv_wave", "head_mlp")) for k in plan.ops), "fp32 MFMA kernel in the bf16 plan"
    rep = _bf16_vs_oracle(net, plan, x, outs, list(range(B)), crop)
    # What an fp32 last class layer would buy (VERDICT r4 #2 "consider keeping the cls head's last 1x1 and the softmax in fp32"):
    # recompute cls.6 in fp32 (torch, fp32 weights) from the engine's bf16 hidden map and count the top-1 decisions that still differ
    # from the free-running oracle's.  The softmax / fg_prob / top-1 already run in fp32 on fp32 logits (m3d_anchor_select).
    sd = synth.synth_state_dict(0)
    c2 = [op for op in plan.ops if op[0] == "cls.6"]           # (the fused cls.3 + cls.6 launch of round 5 keeps the hidden map in LDS: the
    fh, fw = crop[0] // 8, crop[1] // 8                         # experiment runs on the unfused plan, M3D_BF16_HEADS2=0; measured: 182 of 199)
    hidden = None                        # the plan-owned bf16 buffer cls.6 reads (256 channels per pixel)
    for t in plan.keep:
        if c2 and torch.is_tensor(t) and t.data_ptr() == c2[0][4].inp:
            hidden = t
    if hidden is not None:
        h = hidden.view(B, fh, fw, 256).float().permute(0, 3, 1, 2)
        logits = F.conv2d(h, sd["cls.6.weight"].to(_dev()).float(), sd["cls.6.bias"].to(_dev()).float())
        A = eng.A
        pr = torch.softmax(logits.reshape(B, 4, A * fh, fw), dim=1)
        fg32 = (1 - pr[:, 0]).view(B, A, fh, fw)
        ind32 = fg32.max(dim=1, keepdim=True)[1].cpu()
        from oracle import model_cpu
        cconf = synth.synth_conf(crop, 0, batch_size=B, device="cpu")
        tf = {}
        with torch.no_grad():
            model_cpu.rpn_forward(sd, cconf, x, tf)
        o_ind = tf["fg_prob"].max(dim=1, keepdim=True)[1]
        rep["n_idx_with_fp32_cls6"] = int((o_ind != ind32).sum())
    _log("bf16_network", rep)
    _assert_bf16_report(rep, full_size=(crop == (384, 1280)))


def test_bf16_configs2_batch64_matches_fp32_oracle_on_a_slice():
    """BASELINE.json configs[2] AT ITS OWN SIZE (bs = 64, 1280x384, the plan the bench times): frames 0, 21, 42, 63 of the batch
    against the fp32 CPU oracle run on those four frames (engine decisions injected), same per-column bounds as the 2-frame test."""
    crop, B = (384, 1280), 64
    net, conf = _net(crop, B, "bf16")
    x = synth.synth_frames(B, crop, 1234)
    rows = [0, 21, 42, 63]
    with torch.no_grad():
        outs = [t[rows].cpu() for t in net(x.to(_dev()))[:4]]
    plan = net.engine().plan_for(B, *crop)
    rep = _bf16_vs_oracle(net, plan, x[rows], outs, rows, crop)
    rep["batch"] = B
    _log("bf16_configs2_slice", rep)
    _assert_bf16_report(rep, full_size=True)


# What the bf16 path means for the DECODED boxes (pixels / metres / radians), measured on the rows the fp32 oracle keeps: the
# normalised regression outputs above go through exp() for the sizes and are scaled by anchor sizes for the centres.
# Measured (2 frames of 1280x384, synthetic weights; gpurun_out/parity_r04.jsonl "bf16_physical_units"): rows kept after NMS --
# 2-D corners median 1.5 px / max 5.4 px (boxes are hundreds of pixels wide: exp() of the size regression), projected 3-D centre
# 0.31 / 0.82 px, depth 0.9 / 2.8 mm, w / h / l 1.1 % / 2.7 %, rotation 0.0025 / 0.007 rad; over the 3000 pre-NMS rows the
# maxima are 21.7 px, 1.8 px, 9 mm, 6.3 %, 0.058 rad.  The bounds below are (median, max) with a 2x margin.
BF16_PHYS_TOL = {"box2d_px": (3.0, 45.0), "xy3d_px": (1.0, 4.0), "z_m": (0.005, 0.02), "whl_rel": (0.03, 0.13), "ry_rad": (0.006, 0.12)}


@pytest.mark.parametrize("crop,B", [((384, 1280), 2)])
def test_bf16_decoded_boxes_in_physical_units(crop, B):
    """The bf16 engine's outputs decoded with the reference's rules (lib/rpn_util.py:1442-1521, oracle/detect.py) next to the fp32
    oracle's (the engine's top-1 anchor / hard-mask decisions injected), on the rows the ORACLE's detector keeps after NMS (<= 40
    per image) and on its 3000 pre-NMS rows: 2-D box corners in pixels, projected 3-D centre in pixels, depth in metres,
    w / h / l relative, rotation in radians.  Asserted as (median, max) pairs over the kept rows."""
    from oracle import detect as odet
    from oracle import model_cpu
    net, conf = _net(crop, B, "bf16")
    sd = synth.synth_state_dict(0)
    x = synth.synth_frames(B, crop, 1234)
    with torch.no_grad():
        cls, prob, b2, b3, fs, rois = (t.cpu() for t in net(x.to(_dev())))
    plan = net.engine().plan_for(B, *crop)
    fh, fw = crop[0] // 8, crop[1] // 8
    ind = plan.named["sel_idx"].view(B, 1, fh, fw).long().cpu()
    prob_sel = plan.named["sel_prob"].view(B, 1, fh, fw).cpu()
    cconf = synth.synth_conf(crop, 0, batch_size=B, device="cpu")
    with torch.no_grad():
        o = model_cpu.rpn_forward(sd, cconf, x, inject={"sel": {"ind": ind, "hard": (prob_sel > 0.5).float()}})
    rep = {}
    for scope in ("kept", "top3000"):
        d = {k: [] for k in BF16_PHYS_TOL}
        dscore = []
        for i in range(B):
            ab, keep, top = odet.detect_image(o[1][i], o[2][i], o[3][i], o[5], cconf)
            rows = torch.from_numpy(top[keep] if scope == "kept" else top)
            r2, r3, rs, _, _ = odet.decode(o[1][i], o[2][i], o[3][i], o[5], cconf)
            g2, g3, gs, _, _ = odet.decode(prob[i], b2[i], b3[i], rois, cconf)
            d["box2d_px"].append((g2[rows] - r2[rows]).abs().max(1)[0])
            d["xy3d_px"].append((g3[rows, :2] - r3[rows, :2]).abs().max(1)[0])
            d["z_m"].append((g3[rows, 2] - r3[rows, 2]).abs())
            d["whl_rel"].append(((g3[rows, 3:6] - r3[rows, 3:6]).abs() / r3[rows, 3:6].abs()).max(1)[0])
            d["ry_rad"].append((g3[rows, 6] - r3[rows, 6]).abs())
            dscore.append((gs[rows] - rs[rows]).abs())
        for k, v in d.items():
            v = torch.cat(v)
            rep["%s_%s" % (scope, k)] = [v.median().item(), v.max().item()]
        rep[scope + "_score"] = torch.cat(dscore).max().item()
        rep[scope + "_rows"] = int(sum(len(v) for v in d["z_m"]))
    _log("bf16_physical_units", rep)
    for k, (tmed, tmax) in BF16_PHYS_TOL.items():
        for scope in ("kept", "top3000"):
            med, mx = rep["%s_%s" % (scope, k)]
            assert med < tmed and mx < tmax, (scope, k, med, mx)
    assert rep["kept_score"] < BF16_PROB_TOL


@pytest.mark.parametrize("B,H,W", [(2, 16, 32), (2, 48, 160)])
def test_anab_pool_nested_bf16_matches_fp32_kernel(B, H, W):
    """m3d_anab_pool_nested_bf16[_ex] (K|V map stored as bf16) against m3d_anab_pool_nested on the widened values: the same fp32 sums
    in the same order (bit-identical), and the bf16 twins of the _ex entry == the rounded fp32 outputs."""
    from m3dssd_amd import _hip
    L, dev = _hip.lib(), _dev()
    ck, cv = 168, 128
    g = torch.Generator().manual_seed(5)
    kv = _r(torch.randn(B * H * W, ck + cv, generator=g))
    sg = torch.rand(B * H * W, 4, generator=g)
    kp, ckp = 384, 192
    outs = []
    for use16 in (False, True):
        khat = torch.zeros(B * kp * ckp, device=dev)
        vhat = torch.zeros(B * cv * kp, device=dev)
        scratch = torch.empty(L.m3d_anab_pool_nested_scratch_bytes(B, ck + cv) // 4, device=dev)
        d_s = sg.to(dev).contiguous()
        if use16:
            d_kv = kv.to(BF16).to(dev).contiguous()
            k16, v16 = torch.zeros(B * kp * ckp, device=dev, dtype=BF16), torch.zeros(B * cv * kp, device=dev, dtype=BF16)
            _hip.check(L.m3d_anab_pool_nested_bf16_ex(d_kv.data_ptr(), ck + cv, d_s.data_ptr(), 4, B, H, W, ck, cv, scratch.data_ptr(),
                                                      khat.data_ptr(), kp, ckp, vhat.data_ptr(), 0, k16.data_ptr(), v16.data_ptr(), _st()))
            torch.cuda.synchronize()
            assert torch.equal(k16, khat.to(BF16)) and torch.equal(v16, vhat.to(BF16))
            k2, v2 = torch.zeros_like(khat), torch.zeros_like(vhat)
            _hip.check(L.m3d_anab_pool_nested_bf16(d_kv.data_ptr(), ck + cv, d_s.data_ptr(), 4, B, H, W, ck, cv, scratch.data_ptr(),
                                                   k2.data_ptr(), kp, ckp, v2.data_ptr(), 0, _st()))
            torch.cuda.synchronize()
            assert torch.equal(k2, khat) and torch.equal(v2, vhat)
        else:
            d_kv = kv.to(dev).contiguous()
            _hip.check(L.m3d_anab_pool_nested(d_kv.data_ptr(), ck + cv, d_s.data_ptr(), 4, B, H, W, ck, cv, scratch.data_ptr(),
                                              khat.data_ptr(), kp, ckp, vhat.data_ptr(), 0, _st()))
        torch.cuda.synchronize()
        outs.append((khat.cpu(), vhat.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][0].abs().max() > 0


def test_bf16_engine_alternative_paths_agree(monkeypatch):
    """The A/B forms of the bf16 engine (three-launch attention, fp32 K|V map, grouped head GEMMs, separate stem / level0 / level1
    launches) compute the same network: head outputs agree with
    the default plan within the rounding differences of the attention block (bf16 probabilities / features)."""
    from m3dssd_amd import engine_bf16
    crop, B = (128, 320), 2
    x = synth.synth_frames(B, crop, 99).to(_dev())
    ref = None
    monkeypatch.setattr(engine_bf16, "TREE_ENTRY", True)
    # (fused ANAB, bf16 K|V, fused heads, fused front end, round-5 heads, fused tree entry)
    for fused, kv16, heads, front, heads2, front2 in [(True, True, True, True, True, True), (False, True, True, True, True, True),
                                                      (True, False, True, True, True, True), (False, False, True, True, True, True),
                                                      (True, True, False, True, True, True), (True, True, True, False, True, True),
                                                      (True, True, True, True, False, True), (True, True, True, True, True, False)]:
        monkeypatch.setattr(engine_bf16, "FUSED_ANAB", fused)
        monkeypatch.setattr(engine_bf16, "KV_BF16", kv16)
        monkeypatch.setattr(engine_bf16, "FUSED_HEADS", heads)
        monkeypatch.setattr(engine_bf16, "FUSED_FRONT", front)
        monkeypatch.setattr(engine_bf16, "HEADS2", heads2)
        monkeypatch.setattr(engine_bf16, "TREE_ENTRY", front2)
        net, _ = _net(crop, B, "bf16")
        with torch.no_grad():
            out = [t.float().cpu() for t in net(x)[:4]]
        kinds = {op[1] for op in net.engine().plan_for(B, *crop).ops}
        assert ("bf16_anab" in kinds) == fused
        assert ("bf16_head2" in kinds) == (heads and heads2) and ("bf16_head_mlp" in kinds) == (heads and not heads2)
        assert ("bf16_frontend2" in kinds) == front and "bf16_frontend" not in kinds
        if ref is None:
            ref = out
            continue
        for name, a, b in zip(("cls", "prob", "bbox_2d", "bbox_3d"), ref, out):
            d = (a - b).abs()
            # different bf16 rounding points, same network: the stated bf16 tolerances (rms / 99.9 % -- doubled, both sides carry the
            # error -- / max; a flipped discrete
            # decision -- top-1 anchor, hard mask -- moves a few entries by more than the rounding noise)
            assert float((d ** 2).mean().sqrt()) <= BF16_BBOX_RMS_TOL and float(torch.quantile(d.flatten()[:2000000], 0.999)) <= 2 * BF16_BBOX_P999_TOL \
                and float(d.max()) <= BF16_BBOX_MAX_TOL, (fused, kv16, heads, front, heads2, front2, name, float(d.max()))


def test_bf16_detections_scored_in_kitti_ap_units_against_the_fp32_detector():
    """a statement about the bf16 path in the units BASELINE.json configs[4] is quoted in, without KITTI.  64 synthetic
    1280x384 frames run through forward -> decode -> NMS -> 3-D refinement -> KITTI result text in fp32 and in bf16
    (tools/bf16_ap_agreement.py); the fp32 rows (shifted by 1 mm: the reference's rotated IoU is 0 for many boxes against a
    bit-identical copy, tests/test_kitti_eval.py) are the pseudo ground truth, `get_official_eval_result` (lib/eval/eval.py:638-747)
    scores both.  fp32 against itself reads ~100 by construction.  Measured for bf16 (round 6, gpurun_out/parity_r06.jsonl
    "bf16_ap_agreement"; moderate, AP_R40): Car image 85.8 / BEV 65.1 / 3-D 58.2, Pedestrian 77.1 / 57.7 / 48.9, Cyclist 84.6 / 81.9 /
    58.5 at IoU 0.7 / 0.5 / 0.5 with 943 fp32 and 946 bf16 rows -- random-init weights, where 1.3-1.9 % of the pixels pick another
    top-1 anchor in bf16 and every such flip moves a box by more than the IoU margin.  NOT AP3D parity (no dataset, no trained
    weights); floors at ~0.75 x measured."""
    from tools import bf16_ap_agreement as T
    r = T.run(64, 8)
    f, b = r["f32"], r["bf16"]
    _log("bf16_ap_agreement", {"rows_f32": r["rows_f32"], "rows_bf16": r["rows_bf16"], "gt_per_class": r["gt_per_class"],
                               "f32": {k: v for k, v in f.items() if "moderate_R40" in k},
                               "bf16": {k: v for k, v in b.items() if "moderate_R40" in k}})
    assert r["rows_f32"] > 500 and abs(r["rows_bf16"] - r["rows_f32"]) <= 0.03 * r["rows_f32"]
    assert min(r["gt_per_class"].values()) >= 40
    for k, v in f.items():                                   # the detector against its own (1 mm shifted) output
        assert v >= 99.0, (k, v)
    floors = {"Car_image_moderate_R40": 64.0, "Car_bev_moderate_R40": 48.0, "Car_3d_moderate_R40": 43.0,
              "Pedestrian_image_moderate_R40": 57.0, "Pedestrian_bev_moderate_R40": 43.0, "Pedestrian_3d_moderate_R40": 36.0,
              "Cyclist_image_moderate_R40": 63.0, "Cyclist_bev_moderate_R40": 61.0, "Cyclist_3d_moderate_R40": 43.0}
    for k, lo in floors.items():
        assert b[k] >= lo, (k, b[k], lo)
        assert b[k] <= f[k] + 1e-6


def test_bf16_batch64_full_size_properties():
    """BASELINE.json configs[2] at its own size: 64 frames of 1280x384.  Size-independent properties: finite outputs, run-to-run
    bit determinism, image i of the batch == the same image in a batch of 2 (bf16 kernels are batch-invariant: same tiles,
    same accumulation order), detections produced for every image."""
    from lib.rpn_util import detect_batch
    net, conf = _net((384, 1280), 64, "bf16")
    dev = _dev()
    x = synth.synth_frames(64, (384, 1280), 7).to(dev)
    with torch.no_grad():
        a = [t.clone() for t in net(x)[:4]]
        b = [t.clone() for t in net(x)[:4]]
        two = [t.clone() for t in net(x[40:42])[:4]]
        dets, counts = (t.clone() for t in detect_batch(net, x, conf))
    for u, v in zip(a, b):
        assert torch.isfinite(u).all() and torch.equal(u, v)
    for name, u, s_ in zip(("cls", "prob", "bbox_2d", "bbox_3d"), a, two):
        assert torch.equal(u[40:42], s_), name
    assert dets.shape == (64, conf.nms_topN_post, 14) and torch.isfinite(dets).all()
    assert (counts > 0).all()


def test_bf16_and_fp32_engines_agree_on_detections():
    """End to end at 1280x384, both engines on the same frames through decode -> top-3000 -> NMS -> top-40."""
    from lib.rpn_util import detect_batch
    dev = _dev()
    x = synth.synth_frames(4, (384, 1280), 55).to(dev)
    out = {}
    for dt in ("f32", "bf16"):
        net, conf = _net((384, 1280), 4, dt)
        d, c = detect_batch(net, x, conf)
        out[dt] = (d.clone().cpu(), c.clone().cpu())
    (d32, c32), (d16, c16) = out["f32"], out["bf16"]
    # The synthetic network's top scores are a near-flat field (30 000 rows above 0.5 per frame), so WHICH anchors survive the
    # top-3000 cut is decided inside the bf16 noise; what must hold is that the kept detections are equally good: the sorted
    # score profile of the kept rows agrees within the probability tolerance, and every image yields detections.
    assert (c16 > 0).all() and (c32 > 0).all()
    worst = 0.0
    for i in range(4):
        k = min(int(c32[i]), int(c16[i]))
        worst = max(worst, (d32[i, :k, 4] - d16[i, :k, 4]).abs().max().item())
    _log("bf16_vs_fp32_detections", dict(score_profile_linf=worst, counts32=c32.tolist(), counts16=c16.tolist()))
    assert worst < BF16_PROB_TOL
