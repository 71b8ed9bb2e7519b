"""bf16 inference engine (BASELINE.json configs[2]: bs = 64, bf16 storage, MFMA v_mfma_f32_32x32x16_bf16).

Same graph and the same plan / launch machinery as ``engine.Engine`` (model/M3d_inference_align.py:215-313), with

  * activations and weights stored as bf16 in HBM, fp32 accumulation, fp32 folded-BatchNorm / bias / residual / activation
    epilogues inside the producing kernel (``m3d_conv_bf16_forward``) -- every activation crosses HBM once, as bf16;
  * everything that decides something discrete or addresses memory stays fp32: DCN offsets / masks, the class logits
    and box regressions written by the last head layers (planar fp32, consumed unchanged by ``m3d_anchor_select``,
    ``m3d_align_offsets``, ``m3d_bundle_outputs`` and the detection stage), the ANAB keys / values before pooling and
    the attention logits before the softmax;
  * the three layers of the RPN heads that read the same feature map run as grouped launches (layer 1 is ONE GEMM
    128 -> G*256, layers 2 / 3 are ``groups = G`` launches).

Numerics contract: bf16 has an 8-bit significand (relative rounding 2^-9); through ~50 layers the 3-D box parameters agree
with the fp32 oracle to ~1e-1 absolute (measured and asserted in tests/test_gpu_bf16.py), against 1e-3 for the fp32 path.
"""
import ctypes
import os

import numpy as np
import torch

from . import _hip
from ._hip import ConvBf16Desc, Head2Bf16Desc, HeadBf16Desc, QkvsBf16Desc, Tail2Bf16Desc, TreeEntryBf16Desc
from .engine import BN_EPS, PSP_SIZES, SELECT_KEYS, Engine, OpCost, _Plan, _rup

BF16 = torch.bfloat16
FUSED_ANAB = os.environ.get("M3D_BF16_FUSED_ANAB", "1") != "0"
KV_BF16 = os.environ.get("M3D_BF16_KV_BF16", "1") != "0"
FUSED_HEADS = os.environ.get("M3D_BF16_FUSED_HEADS", "1") != "0"
USE_WIDE = os.environ.get("M3D_BF16_WIDE", "1") != "0"       # 3x3 layers on the 128 x 128 wave-tile kernel where it applies
FUSED_FRONT = os.environ.get("M3D_BF16_FUSED_FRONT", "1") != "0"
BRANCH = os.environ.get("M3D_BF16_BRANCH", "1") != "0"         # ANAB + z3d head as a side branch beside the size heads
TREE_ENTRY = os.environ.get("M3D_BF16_TREE_ENTRY", "1") != "0"   # max-pool + project + stride-2 conv1 of a tree in one launch (csrc/bf16_tree_entry.hip)
HEADS2 = os.environ.get("M3D_BF16_HEADS2", "1") != "0"         # round-5 form of the fused heads (csrc/bf16_head_mlp2.hip)
SHAPE_PATCH = os.environ.get("M3D_BF16_SHAPE_PATCH", "0") != "0"   # shape_align through the LDS-patch DCNv2 kernel (per-tile decision)



def _f16_checked(t, what):
    """fp16 copy of folded weights for the kernels that compute in fp16 inside a launch; a value outside fp16's range has no
    faithful copy: refuse (the round-4 kernels of the same layer take bf16 weights: the switch named in the message)."""
    t = t.float()
    if not bool(torch.isfinite(t).all()) or float(t.abs().max()) > 65504.0:
        raise RuntimeError("bf16 engine: %s with the BatchNorm scale folded in leaves the fp16 range (|w| max %g)" % (what, float(t.abs().max())))
    return t.to(torch.float16)


def pack_frontend_f16(w_stem, bn_stem, w_l0, bn_l0, w_l1, bn_l1, device):
    """Operands of m3d_frontend2_bf16_forward (csrc/bf16_frontend2.hip): fp16 weights with the folded BatchNorm scale multiplied
    in, fp32 shifts.  bn_* = (scale [C], shift [C]).
    stem: A fragments of v_mfma_f32_32x32x16_f16 [7 tap rows][2 K-steps][64 lanes][8]: MFMA row r = lane % 32 is channel
    4 * (r / 8) + r % 4 of output pixel `shift` = (r % 8 >= 4) of a pixel pair, k = 8 * (lane / 32) + e is window column
    4 * kstep + 2 * (lane / 32) + e / 4, colour e % 4; the weight is w[ch, colour, i, column - shift]; colour slot 3 (the kernel
    writes 1.0 there) carries the folded BatchNorm SHIFT at tap (0, 0) of the pixel.
    level0 / level1: [Co][160], k = (i * 3 + j) * 16 + c, zero past 144."""
    ws = (w_stem.detach().float().cpu() * bn_stem[0].detach().float().cpu()[:, None, None, None])    # [16, 3, 7, 7]
    frag = torch.zeros(7, 2, 64, 8)
    for lane in range(64):
        r, kh = lane % 32, lane // 32
        shift = 1 if (r % 8) >= 4 else 0
        ch = 4 * (r // 8) + r % 4
        for s in range(2):
            for e in range(8):
                col, rgb = 4 * s + 2 * kh + e // 4, e % 4
                j = col - shift
                if rgb < 3 and 0 <= j < 7:
                    frag[:, s, lane, e] = ws[ch, rgb, :, j]
                elif rgb == 3 and j == 0:
                    frag[0, s, lane, e] = bn_stem[1][ch]       # the shift: slot 3 of every image-tile pixel holds 1.0
    out = [_f16_checked(frag, "front end stem weights").to(device).contiguous(), bn_stem[1].detach().float().to(device).contiguous()]
    for w, bn in ((w_l0, bn_l0), (w_l1, bn_l1)):
        co = w.shape[0]
        wf = w.detach().float().cpu() * bn[0].detach().float().cpu()[:, None, None, None]
        t = torch.zeros(co, 160)
        t[:, :144] = wf.permute(0, 2, 3, 1).reshape(co, 144)
        out += [_f16_checked(t, "front end level0 / level1 weights").to(device).contiguous(), bn[1].detach().float().to(device).contiguous()]
    return out


def _head2_frag(w, dtype):
    """[256, K] fp32 (scale folded) -> A-operand fragments of the 8 waves of csrc/bf16_head_mlp2.hip: [8][K/16][64][8]; MFMA row
    r = lane % 32 of wave w is channel 32 w + 16 ((r % 8) / 4) + 4 (r / 8) + r % 4, element e is k = 16 s + 8 (lane / 32) + e."""
    K = w.shape[1]
    r = torch.arange(32)
    ch_of_row = 16 * ((r % 8) // 4) + 4 * (r // 8) + r % 4                  # [32]
    lane = torch.arange(64)
    ch = (32 * torch.arange(8)[:, None] + ch_of_row[lane % 32][None, :])     # [8, 64]
    k = (16 * torch.arange(K // 16)[:, None, None] + 8 * (lane // 32)[None, :, None] + torch.arange(8)[None, None, :])   # [K/16, 64, 8]
    out = w[ch[:, None, :, None], k[None, :, :, :]]                          # [8, K/16, 64, 8]
    return (_f16_checked(out, "head weights") if dtype == torch.float16 else out.to(dtype)).contiguous()


def pack_head2(heads, device):
    """Operands of m3d_head_mlp2_bf16_forward for the heads of one launch: heads = [(w1 [256,128], s1, t1, w2 [256,256], s2, t2,
    w3 [Cout,256], s3, t3)] fp32 (s / t = folded BatchNorm / bias scale and shift) -> (w1f bf16, w2f fp16, w3 fp16 [G,64,256],
    t1 [G,256], t2 [G,256], t3 [G,64]) with the scales multiplied into the weights."""
    w1f, w2f, w3p, t1, t2, t3 = [], [], [], [], [], []
    for (w1, s1, b1, w2, s2, b2, w3, s3, b3) in heads:
        f = lambda t: t.detach().float().cpu()                # noqa: E731
        w1f.append(_head2_frag(f(w1).reshape(256, 128) * f(s1)[:, None], BF16))
        w2f.append(_head2_frag(f(w2).reshape(256, 256) * f(s2)[:, None], torch.float16))
        co = w3.shape[0]
        w3s = torch.zeros(64, 256)
        w3s[:co] = f(w3).reshape(co, 256) * f(s3)[:, None]
        w3p.append(_f16_checked(w3s, "head output weights"))
        t1.append(f(b1)); t2.append(f(b2))
        tt = torch.zeros(64)
        tt[:co] = f(b3)
        t3.append(tt)
    return tuple(torch.stack(x).to(device).contiguous() for x in (w1f, w2f, w3p, t1, t2, t3))


def pack_tail2(wa, sa, ta, wb, sb, tb, device):
    """Operands of m3d_head_tail2_bf16_forward: wa [256,256] (+ scale sa, shift ta: 1x1 256 -> 256, LeakyReLU behind it), wb [Cout,256]
    (+ sb, tb: 1x1 256 -> Cout) fp32 -> (waf bf16, wbf fp16 fragments [8][16][64][8], t1 [256], t2 [256]) with the scales folded."""
    f = lambda t: t.detach().float().cpu()                    # noqa: E731
    co = wb.shape[0]
    wbp, t2 = torch.zeros(256, 256), torch.zeros(256)
    wbp[:co] = f(wb).reshape(co, 256) * f(sb)[:, None]
    t2[:co] = f(tb)
    return (_head2_frag(f(wa).reshape(256, 256) * f(sa)[:, None], BF16).to(device), _head2_frag(wbp, torch.float16).to(device),
            f(ta).to(device).contiguous(), t2.to(device).contiguous())


def pack_tree_entry(w1, s1, wp, sp, device):
    """Weights of m3d_tree_entry_bf16_forward: w1 [Cout, Cin, 3, 3] (tree1.conv1, + scale s1), wp [Cout, Cin, 1, 1] (project, + scale
    sp) fp32 -> fp16 fragments [Cout/32][Cin/32][10 taps][2 blocks][64 lanes][8]: row r = lane % 16 of block b of slice ws is channel
    32 ws + 8 (r / 4) + 4 b + r % 4; element e of lane l is input channel 32 c + 8 (l / 16) + e; taps 0..8 = the 3x3 taps row-major,
    tap 9 = the 1x1; scales folded."""
    f = lambda t: t.detach().float().cpu()                    # noqa: E731
    co, ci = w1.shape[0], w1.shape[1]
    wall = torch.cat([f(w1).reshape(co, ci, 9) * f(s1)[:, None, None], f(wp).reshape(co, ci, 1) * f(sp)[:, None, None]], 2)   # [Co, Ci, 10]
    lane = torch.arange(64)
    r, kgl = lane % 16, lane // 16
    ws = torch.arange(co // 32)
    b = torch.arange(2)
    ch = 32 * ws[:, None, None] + 8 * (r // 4)[None, None, :] + 4 * b[None, :, None] + (r % 4)[None, None, :]      # [WS, 2, 64]
    cin = 32 * torch.arange(ci // 32)[:, None, None] + 8 * kgl[None, :, None] + torch.arange(8)[None, None, :]      # [C, 64, 8]
    # out[ws, c, tap, b, lane, e] = wall[ch[ws, b, lane], cin[c, lane, e], tap]
    out = wall[ch[:, None, None, :, :, None], cin[None, :, None, None, :, :], torch.arange(10)[None, None, :, None, None, None]]
    return _f16_checked(out, "tree entry weights").contiguous().to(device)


class View16:
    """NHWC bf16 view (possibly a channel slice) of a device buffer; strides in elements."""
    __slots__ = ("t", "ptr", "n", "h", "w", "c", "cs", "esize")

    def __init__(self, t, n, h, w, c, cs=None, ptr=None):
        self.t, self.n, self.h, self.w, self.c = t, n, h, w, c
        self.cs = c if cs is None else cs
        self.esize = t.element_size()
        self.ptr = t.data_ptr() if ptr is None else ptr

    def slice(self, c0, c):
        assert c0 % 8 == 0 and c0 + c <= self.cs
        return View16(self.t, self.n, self.h, self.w, c, self.cs, self.ptr + self.esize * c0)

    def torch_nchw(self):
        full = self.t.view(self.n, self.h, self.w, self.cs)
        off = (self.ptr - self.t.data_ptr()) // self.esize
        return full[..., off:off + self.c].permute(0, 3, 1, 2).float().contiguous()


def pack_conv_bf16(weight, cout_pad=None, cin_pad=None, device=None):
    """[Cout, Cin, kh, kw] -> bf16 [Cout_pad][Kpad], K = (i*kw + j)*Cin_pad + c (tap-major), zero padded; Kpad % 64 == 0."""
    w = weight.detach().to(device if device is not None else weight.device, torch.float32)
    co, ci, kh, kw = w.shape
    cip = ci if cin_pad is None else cin_pad
    cop = _rup(co, 32) if cout_pad is None else cout_pad
    k = kh * kw * cip
    kpad = _rup(k, 64)
    p = torch.zeros(cop, kh * kw, cip, device=w.device, dtype=torch.float32)
    p[:co, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, kh * kw, ci)
    out = torch.zeros(cop, kpad, device=w.device, dtype=BF16)
    out[:, :k] = p.reshape(cop, k).to(BF16)
    return out.contiguous(), kpad


class PackedBf16:
    def f16(self):
        """fp16 copy of the packed weights for the LDS-patch DCNv2 kernel (exact for |w| in [6.1e-5, 65504]; made on first use).
        None when a weight exceeds the fp16 range (it would become inf): the layer then stays on the implicit-GEMM kernel."""
        if getattr(self, "_wp16", None) is None:
            w = self.wp.float()
            if not bool(torch.isfinite(w).all()) or float(w.abs().max()) > 65504.0:
                self._wp16 = False
            else:
                self._wp16 = w.to(torch.float16).contiguous()
        return self._wp16 if self._wp16 is not False else None

    def wave3x3(self):
        """The 3x3 weights in the fragment order of csrc/bf16_conv_wide.hip: [Cout_pad/128][Cin/32][9 taps][2 K-steps of 16][4 blocks of
        32 channels][64 lanes][8], lane = 32 * ((c % 16) / 8) + (channel % 32); None where the kernel does not apply."""
        if getattr(self, "_wave", None) is None:
            if self.kh != 3 or self.kw != 3 or self.cin % 32 or self.cout_pad % 128:
                return None
            w = self.wp[:, :9 * self.cin].reshape(self.cout_pad // 128, 4, 32, 9, self.cin // 32, 2, 2, 8)   # g, ct, r, tap, ch, ks, h, e
            self._wave = w.permute(0, 4, 3, 5, 1, 6, 2, 7).contiguous()
        return self._wave

    def __init__(self, eng, weight, bias=None, bn=None, cout_pad=None):
        dev = eng.device
        self.cout, self.cin, self.kh, self.kw = weight.shape
        self.wp, self.kpad = pack_conv_bf16(weight, cout_pad, None, dev)
        self.cout_pad = self.wp.shape[0]
        scale = torch.ones(self.cout, device=dev, dtype=torch.float32)
        shift = torch.zeros(self.cout, device=dev, dtype=torch.float32)
        if bias is not None:
            shift = bias.detach().to(dev, torch.float32).clone()
        if bn is not None:
            g, b, m, v = (t.detach().to(dev, torch.float32) for t in bn)
            s = g / torch.sqrt(v + BN_EPS)
            shift = (shift - m) * s + b
            scale = s
        self.scale, self.shift = scale.contiguous(), shift.contiguous()
        self.has_affine = (bias is not None) or (bn is not None)


class EngineBF16(Engine):
    compute_dtype = "bf16"

    # ------------------------------------------------------------------ parameters
    def _pc(self, conv, bn=None, **kw):
        sd = self.sd
        return PackedBf16(self, sd[conv + ".weight"], sd.get(conv + ".bias"), self._bn(bn) if bn else None, **kw)

    def _pack(self, sd):
        dev = self.device
        P = {}
        b = "base.base"
        w = sd[b + ".base_layer.0.weight"].detach().to(dev, torch.float32)
        P["stem.w"] = w.permute(2, 3, 1, 0).contiguous()
        g, be, m, v = (t.detach().to(dev, torch.float32) for t in self._bn(b + ".base_layer.1"))
        s = g / torch.sqrt(v + BN_EPS)
        P["stem.scale"], P["stem.shift"] = s.contiguous(), (be - m * s).contiguous()
        P["level0"] = self._pc(b + ".level0.0", b + ".level0.1")
        P["level1"] = self._pc(b + ".level1.0", b + ".level1.1")
        P["front2"] = pack_frontend_f16(sd[b + ".base_layer.0.weight"], (P["stem.scale"], P["stem.shift"]),
                                        sd[b + ".level0.0.weight"], (P["level0"].scale, P["level0"].shift),
                                        sd[b + ".level1.0.weight"], (P["level1"].scale, P["level1"].shift), dev)

        def block(p):
            P[p + ".conv1"] = self._pc(p + ".conv1", p + ".bn1")
            P[p + ".conv2"] = self._pc(p + ".conv2", p + ".bn2")

        def tree1(p):
            block(p + ".tree1")
            block(p + ".tree2")
            P[p + ".root"] = self._pc(p + ".root.conv", p + ".root.bn")
            if (p + ".project.0.weight") in sd:
                P[p + ".project"] = self._pc(p + ".project.0", p + ".project.1")

        tree1(b + ".level2")
        for lv in (3, 4):
            tree1("%s.level%d.tree1" % (b, lv))
            tree1("%s.level%d.tree2" % (b, lv))
        tree1(b + ".level5")

        def deform(p):
            P[p + ".om"] = self._pc(p + ".conv.conv_offset_mask")
            P[p + ".dcn"] = PackedBf16(self, sd[p + ".conv.weight"], sd[p + ".conv.bias"], self._bn(p + ".actf.0"))

        def ida(p, n):
            for i in range(1, n):
                deform("%s.proj_%d" % (p, i))
                deform("%s.node_%d" % (p, i))
                up = sd["%s.up_%d.weight" % (p, i)].detach().to(dev, torch.float32)
                P["%s.up_%d" % (p, i)] = up[:, 0].permute(1, 2, 0).contiguous()

        ida("base.dla_up.ida_0", 2)
        ida("base.dla_up.ida_1", 3)
        ida("base.ida_up", 2)
        self.P = P
        if not self.backbone_only:
            self._pack_heads_bf16(sd, P)
        torch.cuda.synchronize(self.device)

    def _pack_heads_bf16(self, sd, P):
        dev = self.device
        self.box_heads = ["bbox_x", "bbox_y", "bbox_w", "bbox_h", "bbox_x3d", "bbox_y3d", "bbox_z3d", "bbox_w3d",
                          "bbox_h3d", "bbox_l3d", "bbox_rY3d"]
        for p in ["cls"] + self.box_heads:
            P[p + ".0"] = self._pc(p + ".0", p + ".1")
            P[p + ".3"] = self._pc(p + ".3", p + ".4")
            P[p + ".6"] = self._pc(p + ".6", cout_pad=_rup(sd[p + ".6.weight"].shape[0], 64))
        for p in ("shape_align", "center_align2d", "center_align3d"):
            P[p] = PackedBf16(self, sd[p + ".align.weight"], sd[p + ".align.bias"], None)
        a = "bbox_z3d_gl.0"
        wq, wk, wv, ws = (sd[a + n].detach().cpu().float() for n in
                          (".query_conv.weight", ".key_conv.weight", ".value_conv.weight", ".spatial_conv.weight"))
        self.ck, self.cv, self.ns = wq.shape[0], wv.shape[0], ws.shape[0]
        self.ck_pad = _rup(self.ck, 64)                                  # K of the logits GEMM
        P["anab.q"] = PackedBf16(self, wq, None, None, cout_pad=self.ck_pad)
        P["anab.kvs"] = PackedBf16(self, torch.cat([wk, wv, ws], 0), None, None)
        P["anab.kv"] = PackedBf16(self, torch.cat([wk, wv], 0), None, None)          # bf16 K|V map + fp32 gates (KV_BF16)
        P["anab.s"] = PackedBf16(self, ws, None, None)
        g, be, m, v = (t.detach().to(dev, torch.float32) for t in self._bn("bbox_z3d_gl.1"))
        s = g / torch.sqrt(v + BN_EPS)
        P["anab.bn.scale"], P["anab.bn.shift"] = s.contiguous(), (be - m * s).contiguous()
        anchors = torch.as_tensor(np.asarray(self.conf.anchors), dtype=torch.float32)
        aw = (anchors[:, 2] - anchors[:, 0])
        ah = (anchors[:, 3] - anchors[:, 1])
        tab = torch.zeros(self.A, 18, dtype=torch.float32)
        h_step, w_step = ah / self.stride / 3, aw / self.stride / 3
        for i in range(3):
            for j in range(3):
                k = i * 3 + j
                tab[:, 2 * k] = (h_step - 1) * (i - 3 / 2 + 0.5)
                tab[:, 2 * k + 1] = (w_step - 1) * (j - 3 / 2 + 0.5)
        P["shape.table"] = tab.to(dev).contiguous()
        P["anchor_wh"] = torch.stack([aw / self.stride, ah / self.stride], 1).to(dev).contiguous()
        P["anchors"] = anchors.to(dev).contiguous()
        P["means"] = torch.as_tensor(np.asarray(self.conf.bbox_means), dtype=torch.float32).reshape(-1).to(dev)
        P["stds"] = torch.as_tensor(np.asarray(self.conf.bbox_stds), dtype=torch.float32).reshape(-1).to(dev)

    # ------------------------------------------------------------------ launch helpers
    def _buf16(self, plan, n, h, w, c, cs=None, name=None, dtype=BF16, zero=False):
        cs = c if cs is None else cs
        t = (torch.zeros if zero else torch.empty)(n * h * w * cs, device=self.device, dtype=dtype)
        plan.keep.append(t)
        v = View16(t, n, h, w, c, cs)
        if name:
            plan.named[name] = v
        return v

    def _conv16(self, plan, name, x, out=None, wgt=None, kpad=None, cout=None, cout_pad=None, kh=1, kw=1, stride=1, pad=0,
                scale=None, shift=None, act=0, res=None, res_mode=0, sigmoid_from=-1, om=None, out_mode=0, planar=None,
                wgt_img_stride=0, groups=1, in_goff=0, wgt_goff=0, out_goff=0, ss_goff=0, cin=None, flops_cin=None, wgt_f16=None, wgt_wave=None):
        """One m3d_conv_bf16_forward launch appended to the plan.  x: View16 (bf16); out: View16 (bf16 or fp32 NHWC) or
        planar = (tensor, img_stride, channel offset) for the fp32 planar staging of the head outputs."""
        d = ConvBf16Desc()
        cin = x.c if cin is None else cin
        d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.ptr, x.cs, x.n, x.h, x.w, cin
        d.wgt, d.wgt_img_stride = wgt.data_ptr() if hasattr(wgt, "data_ptr") else wgt, wgt_img_stride
        d.Cout, d.Cout_pad, d.Kpad = cout, cout_pad, kpad
        d.kh, d.kw, d.stride, d.pad = kh, kw, stride, pad
        d.Ho = (x.h + 2 * pad - kh) // stride + 1
        d.Wo = (x.w + 2 * pad - kw) // stride + 1
        if planar is not None:
            t, img_stride, ch_off = planar
            d.out, d.out_mode, d.out_img_stride = t.data_ptr() + 4 * ch_off * d.Ho * d.Wo, 2, img_stride
        else:
            assert out.h == d.Ho and out.w == d.Wo, (name, out.h, out.w, d.Ho, d.Wo)
            d.out, d.out_cs, d.out_mode = out.ptr, out.cs, out_mode
        if scale is not None:
            d.scale = scale.data_ptr()
        if shift is not None:
            d.shift = shift.data_ptr()
        plan.keep += [t for t in (scale, shift, wgt) if hasattr(t, "data_ptr")]
        if res is not None:
            d.res, d.res_cs, d.res_mode = res.ptr, res.cs, res_mode
        d.act, d.sigmoid_from = act, sigmoid_from
        if om is not None:
            d.dcn_offmask, d.dcn_om_cs = om.ptr, om.cs
            if wgt_f16 is not None:
                # LDS-patch DCNv2 kernel (csrc/bf16_dcn_patch.hip): fp16 copy of the packed weights + one flag word per pixel tile; the
                # library decides per tile, on the device, whether the sampling window fits
                ws = torch.zeros(max(256, self.L.m3d_conv_bf16_dcn_ws_bytes(x.n, out.h, out.w) // 4), device=self.device, dtype=torch.int32)
                plan.keep += [wgt_f16, ws]
                d.wgt_f16, d.dcn_ws, d.dcn_ws_bytes = wgt_f16.data_ptr(), ws.data_ptr(), ws.numel() * 4
        elif wgt_wave is not None:
            plan.keep.append(wgt_wave)       # 128 x 128 wave-tile kernel (csrc/bf16_conv_wide.hip) where the library finds it applicable
            d.wgt_wave = wgt_wave.data_ptr()
        d.groups, d.in_group_off, d.wgt_group_off, d.out_group_off, d.ss_group_off = groups, in_goff, wgt_goff, out_goff, ss_goff
        L = self.L
        ref = ctypes.byref(d)
        flops = 2.0 * x.n * d.Ho * d.Wo * cout * kh * kw * (flops_cin if flops_cin is not None else cin) * groups
        bn = 128 if cout_pad % 128 == 0 else (64 if cout_pad % 64 == 0 else 32)
        variant = L.m3d_conv_bf16_variant(ref)          # which kernel the library runs for this descriptor
        if variant == 5:
            kind = "bf16_wide<128,128>"                           # 3x3 with 128-pixel x 128-channel wave tiles
        elif variant == 8:
            kind = "bf16_c64"                                     # 3x3 64 -> 64, persistent workgroups, weights resident in LDS (level2)
        elif variant == 6:
            kind = "bf16_dcn1x1"                                  # 1x1 DCNv2 128 -> 128 (center_align; csrc/bf16_dcn1x1.hip)
        elif variant >= 3:
            kind = "bf16_dcn_patch<%d>" % (8 * (variant - 2))     # LDS-patch DCNv2 (+ the gated implicit-GEMM fallback behind it)
        elif variant:
            kind = "bf16_halo<%d,%d>" % (bn, 16 * variant)
        else:
            kind = "bf16_conv<%d%s>" % (bn, ",deform" if om is not None else "")
        plan.ops.append((name, kind, flops, lambda st: _hip.check(L.m3d_conv_bf16_forward(ref, st)), d))

    def _pconv(self, plan, name, pc, x, out, stride=1, pad=0, act=0, res=None, res_mode=0, sigmoid_from=-1, om=None, out_mode=0,
               affine=True, patch=True):
        self._conv16(plan, name, x, out, wgt=pc.wp, kpad=pc.kpad, cout=pc.cout, cout_pad=pc.cout_pad, kh=pc.kh, kw=pc.kw,
                     stride=stride, pad=pad, scale=pc.scale if (affine and pc.has_affine) else None,
                     shift=pc.shift if (affine and pc.has_affine) else None, act=act, res=res, res_mode=res_mode,
                     sigmoid_from=sigmoid_from, om=om, out_mode=out_mode, cin=pc.cin,
                     wgt_f16=pc.f16() if (om is not None and patch) else None,
                     wgt_wave=pc.wave3x3() if (USE_WIDE and om is None and stride == 1 and pad == 1 and out_mode == 0
                                               and sigmoid_from < 0 and x.c == pc.cin) else None)

    # ------------------------------------------------------------------ plan construction
    def _build_plan(self, B, H, W):
        L, P = self.L, self.P
        plan = _Plan()
        b = "base.base"
        in_ptr = [0]
        plan.named["input_ptr"] = in_ptr
        in_u8 = [0, 0, 0]
        plan.named["input_u8"] = in_u8
        mean3 = (ctypes.c_float * 3)(*[float(v) for v in self.conf.image_means])
        stds3 = (ctypes.c_float * 3)(*[float(v) for v in self.conf.image_stds])
        l1 = self._buf16(plan, B, H // 2, W // 2, 32, name="level1")
        if FUSED_FRONT:
            # stem -> level0 -> level1 in one launch: the 16-channel full-resolution maps never reach HBM
            f2 = P["front2"]

            def front(st):
                u8 = 1 if in_u8[0] else 0
                _hip.check(L.m3d_frontend2_bf16_forward(
                    in_u8[0] if u8 else in_ptr[0], u8, in_u8[1], in_u8[2], mean3, stds3, f2[0].data_ptr(), f2[1].data_ptr(),
                    f2[2].data_ptr(), f2[3].data_ptr(), f2[4].data_ptr(), f2[5].data_ptr(), l1.ptr, l1.cs, B, H, W, st))
            flops = 2.0 * B * H * W * (147 * 16 + 144 * 16) + 2.0 * B * (H // 2) * (W // 2) * 144 * 32
            # (bytes: the fp32 NCHW image in, the 32-channel half-resolution bf16 map out)
            plan.ops.append(("stem+level0+level1", "bf16_frontend2", flops, front,
                             OpCost(B * H * W * 3 * 4 + B * (H // 2) * (W // 2) * 32 * 2)))
        else:
            s0 = self._buf16(plan, B, H, W, 16)

            def stem(st):
                u8 = 1 if in_u8[0] else 0
                _hip.check(L.m3d_stem_conv7x7_bf16(in_u8[0] if u8 else in_ptr[0], u8, in_u8[1], in_u8[2], mean3, stds3,
                                                   P["stem.w"].data_ptr(), P["stem.scale"].data_ptr(), P["stem.shift"].data_ptr(),
                                                   s0.ptr, s0.cs, B, H, W, st))
            self._op(plan, "stem", "stem_bf16", stem, flops=2.0 * B * H * W * 16 * 147, nbytes=B * H * W * (3 * 4 + 16 * 2))
            l0 = self._buf16(plan, B, H, W, 16, name="level0")
            self._pconv(plan, "level0", P["level0"], s0, l0, 1, 1, act=1)
            self._pconv(plan, "level1", P["level1"], l0, l1, 2, 1, act=1)

        def maxpool(name, x, out):
            self._op(plan, name, "maxpool_bf16", lambda st: _hip.check(L.m3d_maxpool2x2_bf16(
                x.ptr, x.cs, out.ptr, out.cs, x.n, x.h, x.w, x.c, st)), nbytes=x.n * x.h * x.w * x.c * 5 // 2)

        def block(p, x, res, out, stride):
            co = P[p + ".conv1"].cout
            t = self._buf16(plan, B, out.h, out.w, co)
            self._pconv(plan, p + ".conv1", P[p + ".conv1"], x, t, stride, 1, act=1)
            self._pconv(plan, p + ".conv2", P[p + ".conv2"], t, out, 1, 1, act=1, res=res)

        def tree_entry(p, x, co, bottom, res, t):
            """maxpool (-> bottom view or None) + project (-> res) + tree1.conv1 (stride 2, -> t) as ONE launch; False = not applicable."""
            if not (TREE_ENTRY and (p + ".project") in P and x.c % 32 == 0 and co % 64 == 0):
                return False
            key = "tree_entry:" + p
            if key not in P:
                sd = self.sd
                P[key] = pack_tree_entry(sd[p + ".tree1.conv1.weight"], P[p + ".tree1.conv1"].scale, sd[p + ".project.0.weight"],
                                         P[p + ".project"].scale, self.device)
            d = TreeEntryBf16Desc()
            d.inp, d.in_cs, d.N, d.H, d.W, d.Cin, d.Cout = x.ptr, x.cs, x.n, x.h, x.w, x.c, co
            d.wfrag, d.shift1, d.shiftp = P[key].data_ptr(), P[p + ".tree1.conv1"].shift.data_ptr(), P[p + ".project"].shift.data_ptr()
            d.t, d.t_cs, d.res, d.res_cs = t.ptr, t.cs, res.ptr, res.cs
            if bottom is not None:
                d.bottom, d.bottom_cs = bottom.ptr, bottom.cs
            ref = ctypes.byref(d)
            if not L.m3d_tree_entry_bf16_applicable(ref):
                return False
            plan.keep += [P[key], P[p + ".tree1.conv1"].shift, P[p + ".project"].shift]
            ho, wo = x.h // 2, x.w // 2
            flops = 2.0 * x.n * ho * wo * co * x.c * 10
            nbytes = x.n * (x.h * x.w * x.c + ho * wo * (2 * co + (x.c if bottom is not None else 0))) * 2 + co * x.c * 10 * 2
            plan.ops.append((p + ".entry", "bf16_tree_entry", flops,
                             lambda st: _hip.check(L.m3d_tree_entry_bf16_forward(ref, st)), OpCost(nbytes)))
            return True

        def tree1(p, x, co, stride, out, bottom=None):
            h, w = x.h // stride, x.w // stride
            cat = self._buf16(plan, B, h, w, 2 * co)
            x2v, x1v = cat.slice(0, co), cat.slice(co, co)
            if stride == 2 and (p + ".project") in P:
                # fused entry: bottom (only materialised when a root reads it: `bottom` given by the caller), project, conv1
                res = self._buf16(plan, B, h, w, co)
                t = self._buf16(plan, B, h, w, co)
                if tree_entry(p, x, co, bottom, res, t):
                    self._pconv(plan, p + ".tree1.conv2", P[p + ".tree1.conv2"], t, x1v, 1, 1, act=1, res=res)
                    block(p + ".tree2", x1v, x1v, x2v, 1)
                    self._pconv(plan, p + ".root", P[p + ".root"], cat, out, 1, 0, act=1)
                    return
                if bottom is not None:
                    maxpool(p + ".downsample", x, bottom)
            if stride == 1:
                bottom = x
            elif bottom is None:
                bottom = self._buf16(plan, B, h, w, x.c)
                maxpool(p + ".downsample", x, bottom)
            if (p + ".project") in P:
                res = self._buf16(plan, B, h, w, co)
                self._pconv(plan, p + ".project", P[p + ".project"], bottom, res, 1, 0, act=0)
            else:
                res = bottom
            block(p + ".tree1", x, res, x1v, stride)
            block(p + ".tree2", x1v, x1v, x2v, 1)
            self._pconv(plan, p + ".root", P[p + ".root"], cat, out, 1, 0, act=1)

        l2 = self._buf16(plan, B, H // 4, W // 4, 64, name="level2")
        tree1(b + ".level2", l1, 64, 2, l2)

        def tree2(p, x, co, out):
            ci = x.c
            h, w = x.h // 2, x.w // 2
            catb = self._buf16(plan, B, h, w, 2 * co + ci + co)
            bottom = catb.slice(2 * co, ci)
            X1 = catb.slice(2 * co + ci, co)
            tree1(p + ".tree1", x, co, 2, X1, bottom=bottom)      # (writes `bottom` itself: fused entry or its own max-pool launch)
            x2v, x1v = catb.slice(0, co), catb.slice(co, co)
            block(p + ".tree2.tree1", X1, X1, x1v, 1)
            block(p + ".tree2.tree2", x1v, x1v, x2v, 1)
            self._pconv(plan, p + ".tree2.root", P[p + ".tree2.root"], catb, out, 1, 0, act=1)

        l3 = self._buf16(plan, B, H // 8, W // 8, 128, name="level3")
        tree2(b + ".level3", l2, 128, l3)
        l4 = self._buf16(plan, B, H // 16, W // 16, 256, name="level4")
        tree2(b + ".level4", l3, 256, l4)
        l5 = self._buf16(plan, B, H // 32, W // 32, 512, name="level5")
        h5, w5 = H // 32, W // 32
        cat5 = self._buf16(plan, B, h5, w5, 1024 + 256)
        bottom5 = cat5.slice(1024, 256)
        res5 = self._buf16(plan, B, h5, w5, 512)
        t5 = self._buf16(plan, B, h5, w5, 512)
        if tree_entry(b + ".level5", l4, 512, bottom5, res5, t5):
            self._pconv(plan, b + ".level5.tree1.conv2", P[b + ".level5.tree1.conv2"], t5, cat5.slice(512, 512), 1, 1, act=1, res=res5)
        else:
            maxpool(b + ".level5.downsample", l4, bottom5)
            self._pconv(plan, b + ".level5.project", P[b + ".level5.project"], bottom5, res5, 1, 0, act=0)
            block(b + ".level5.tree1", l4, res5, cat5.slice(512, 512), 2)
        block(b + ".level5.tree2", cat5.slice(512, 512), cat5.slice(512, 512), cat5.slice(0, 512), 1)
        self._pconv(plan, b + ".level5.root", P[b + ".level5.root"], cat5, l5, 1, 0, act=1)

        # ---- DLAUp / IDAUp: offsets / masks stay fp32 (they address memory) ---------------------------
        def deform(p, x, out):
            om = self._buf16(plan, B, x.h, x.w, 27, 32, name=p + ".om", dtype=torch.float32)
            self._pconv(plan, p + ".offset_mask", P[p + ".om"], x, om, 1, 1, act=0, sigmoid_from=18, out_mode=1)
            self._pconv(plan, p + ".dcn", P[p + ".dcn"], x, out, 1, 1, act=1, om=om)
            plan.named[p + ".out"] = out

        def ida_step(p, i, x, skip, co):
            proj = self._buf16(plan, B, x.h, x.w, co)
            deform("%s.proj_%d" % (p, i), x, proj)
            summed = self._buf16(plan, B, 2 * x.h, 2 * x.w, co)
            upw = P["%s.up_%d" % (p, i)]
            self._op(plan, "%s.up_%d" % (p, i), "upsample_bf16", lambda st: _hip.check(L.m3d_upsample2x_add_bf16(
                proj.ptr, proj.cs, upw.data_ptr(), skip.ptr, skip.cs, summed.ptr, summed.cs, B, proj.h, proj.w, co, st)),
                flops=2.0 * B * 4 * proj.h * proj.w * co * 4, nbytes=B * proj.h * proj.w * co * 2 * (1 + 4 + 4))
            node = self._buf16(plan, B, 2 * x.h, 2 * x.w, co)
            deform("%s.node_%d" % (p, i), summed, node)
            return node

        L5a = ida_step("base.dla_up.ida_0", 1, l5, l4, 256)
        L4b = ida_step("base.dla_up.ida_1", 1, l4, l3, 128)
        L5b = ida_step("base.dla_up.ida_1", 2, L5a, L4b, 128)
        feats0 = ida_step("base.ida_up", 1, L5a, L5b, 128)
        plan.named["feats0"] = feats0
        plan.feat = (feats0.h, feats0.w)
        if self.backbone_only:
            return plan

        # ---- RPN heads ----------------------------------------------------------------
        fh, fw = feats0.h, feats0.w
        HW = fh * fw
        A, NC = self.A, self.NC
        R = A * HW
        cls_pl = torch.empty(B * NC * A * HW, device=self.device, dtype=torch.float32)
        box_pl = torch.empty(B * 11 * A * HW, device=self.device, dtype=torch.float32)
        plan.keep += [cls_pl, box_pl]
        plan.named["cls_planar"], plan.named["box_planar"] = cls_pl, box_pl
        # every op from here on may write the planar staging: a detection stage that reads it for the PREVIOUS batch
        # (m3dssd_amd.pipeline.PipelinedDetector, planar form) has to be done before op `planar_first_op` of this one starts
        plan.named["planar_first_op"] = len(plan.ops)

        def stacked(names, li):
            key = "stack:" + "+".join(names) + li
            if key not in P:
                P[key] = (torch.cat([P[n + li].wp for n in names], 0).contiguous(),
                          torch.cat([P[n + li].scale for n in names]).contiguous(),
                          torch.cat([P[n + li].shift for n in names]).contiguous())
            return P[key]

        def heads(names, x, first_box_index):
            """The heads `names` (consecutive rows of the planar box staging starting at first_box_index) read the same map x:
            one fused 3-layer launch (m3d_head_mlp_bf16_forward, blockIdx.y = head), hidden activations in LDS.  With
            M3D_BF16_FUSED_HEADS=0: layer 1 as ONE GEMM Cin -> G*256, layers 2 and 3 as grouped launches."""
            G = len(names)
            w1, s1, t1 = stacked(names, ".0")
            w2, s2, t2 = stacked(names, ".3")
            w3, s3, t3 = stacked(names, ".6")
            cp = P[names[0] + ".6"].cout_pad
            if FUSED_HEADS and HEADS2 and x.c == 128 and A <= 64:
                key2 = "head2:" + "+".join(names)
                if key2 not in P:
                    sd = self.sd
                    P[key2] = pack_head2([(sd[n + ".0.weight"], P[n + ".0"].scale, P[n + ".0"].shift,
                                           sd[n + ".3.weight"], P[n + ".3"].scale, P[n + ".3"].shift,
                                           sd[n + ".6.weight"], P[n + ".6"].scale, P[n + ".6"].shift) for n in names], self.device)
                pk = P[key2]
                d = Head2Bf16Desc()
                d.inp, d.in_cs, d.M = x.ptr, x.cs, B * HW
                d.w1f, d.w2f, d.w3, d.t1, d.t2, d.t3 = (t.data_ptr() for t in pk)
                d.Cout = A
                d.out = box_pl.data_ptr() + 4 * first_box_index * A * HW
                d.out_group_off, d.out_img_stride, d.HW, d.groups = A * HW, 11 * A * HW, HW, G
                ref = ctypes.byref(d)
                plan.keep += list(pk)
                flops = 2.0 * B * HW * G * (128 * 256 + 256 * 256 + 256 * A)
                plan.ops.append(("+".join(names) + ".mlp", "bf16_head2", flops,
                                 lambda st: _hip.check(L.m3d_head_mlp2_bf16_forward(ref, st)), d))
                return
            if FUSED_HEADS and x.c == 128 and cp == 64:
                d = HeadBf16Desc()
                d.inp, d.in_cs, d.M, d.Cin = x.ptr, x.cs, B * HW, 128
                d.w1, d.w2, d.w3 = w1.data_ptr(), w2.data_ptr(), w3.data_ptr()
                d.s1, d.t1, d.s2, d.t2, d.s3, d.t3 = (t.data_ptr() for t in (s1, t1, s2, t2, s3, t3))
                d.Cout, d.Cout_pad = A, cp
                d.out = box_pl.data_ptr() + 4 * first_box_index * A * HW
                d.out_group_off, d.out_img_stride, d.HW, d.groups = A * HW, 11 * A * HW, HW, G
                ref = ctypes.byref(d)
                flops = 2.0 * B * HW * G * (128 * 256 + 256 * 256 + 256 * A)
                plan.ops.append(("+".join(names) + ".mlp", "bf16_head_mlp", flops,
                                 lambda st: _hip.check(L.m3d_head_mlp_bf16_forward(ref, st)), d))
                return
            h1 = self._buf16(plan, B, fh, fw, G * 256)
            self._conv16(plan, "+".join(names) + ".0", x, h1, wgt=w1, kpad=P[names[0] + ".0"].kpad, cout=G * 256, cout_pad=G * 256,
                         scale=s1, shift=t1, act=1)
            h2 = self._buf16(plan, B, fh, fw, G * 256)
            self._conv16(plan, "+".join(names) + ".3", h1, h2, wgt=w2, kpad=256, cout=256, cout_pad=256, scale=s2, shift=t2, act=1,
                         groups=G, in_goff=256, wgt_goff=256 * 256, out_goff=256, ss_goff=256, cin=256)
            self._conv16(plan, "+".join(names) + ".6", h2, None, wgt=w3, kpad=256, cout=A, cout_pad=cp, scale=s3, shift=t3,
                         planar=(box_pl, 11 * A * HW, first_box_index * A), groups=G, in_goff=256, wgt_goff=cp * 256,
                         out_goff=A * HW, ss_goff=A, cin=256)

        # cls head: 3x3 128 -> 256, 1x1 256 -> 256, 1x1 256 -> NC*A (planar fp32)
        c1 = self._buf16(plan, B, fh, fw, 256)
        self._pconv(plan, "cls.0", P["cls.0"], feats0, c1, 1, 1, act=1)
        pc = P["cls.6"]
        if FUSED_HEADS and HEADS2 and P["cls.3"].cout == 256 and P["cls.3"].cin == 256 and P["cls.3"].kh == 1 and pc.cout <= 256:
            # cls.3 + cls.6 in one launch (csrc/bf16_head_mlp2.hip: bf16_tail2_kernel), weights resident in registers
            if "cls.tail2" not in P:
                sd = self.sd
                P["cls.tail2"] = pack_tail2(sd["cls.3.weight"], P["cls.3"].scale, P["cls.3"].shift,
                                            sd["cls.6.weight"], pc.scale, pc.shift, self.device)
            pk = P["cls.tail2"]
            d = Tail2Bf16Desc()
            d.inp, d.in_cs, d.M = c1.ptr, c1.cs, B * HW
            d.waf, d.wbf, d.t1, d.t2 = (t.data_ptr() for t in pk)
            d.Cout, d.out, d.out_img_stride, d.HW = pc.cout, cls_pl.data_ptr(), NC * A * HW, HW
            ref = ctypes.byref(d)
            plan.keep += list(pk)
            plan.ops.append(("cls.3+cls.6", "bf16_tail2", 2.0 * B * HW * (256 * 256 + 256 * pc.cout),
                             lambda st: _hip.check(L.m3d_head_tail2_bf16_forward(ref, st)), d))
        else:
            c2 = self._buf16(plan, B, fh, fw, 256)
            self._pconv(plan, "cls.3", P["cls.3"], c1, c2, 1, 0, act=1)
            self._conv16(plan, "cls.6", c2, None, wgt=pc.wp, kpad=pc.kpad, cout=pc.cout, cout_pad=pc.cout_pad, scale=pc.scale,
                         shift=pc.shift, planar=(cls_pl, NC * A * HW, 0), cin=256)
        sel_idx = torch.empty(B * HW, device=self.device, dtype=torch.int32)
        sel_prob = torch.empty(B * HW, device=self.device, dtype=torch.float32)
        plan.keep += [sel_idx, sel_prob]
        plan.named["sel_idx"], plan.named["sel_prob"] = sel_idx, sel_prob
        if NC == 4 and A >= 4 and SELECT_KEYS:
            # + the detection stage's sort keys (plan.named["score_bits"], created below) while the logits are in registers
            plan.named["keys_by_select"] = True
            plan.named["score_bits_first_write_op"] = len(plan.ops)     # a detect(k-1) beside forward(k) must be done before it
            self._op(plan, "anchor_select", "select", lambda st: _hip.check(L.m3d_anchor_select_keys(
                cls_pl.data_ptr(), B, A, HW, sel_idx.data_ptr(), sel_prob.data_ptr(), plan.named["score_bits"].data_ptr(), st)),
                nbytes=B * HW * (A * NC + 2 + A) * 4)
        else:
            self._op(plan, "anchor_select", "select", lambda st: _hip.check(L.m3d_anchor_select(
                cls_pl.data_ptr(), B, A, NC, HW, sel_idx.data_ptr(), sel_prob.data_ptr(), None, st)),
                nbytes=B * HW * (A * NC + 2) * 4)
        means = np.asarray(self.conf.bbox_means, dtype=np.float32).reshape(-1)
        stds = np.asarray(self.conf.bbox_stds, dtype=np.float32).reshape(-1)

        def box_ptr(k):
            return box_pl.data_ptr() + 4 * k * A * HW

        om_sa = self._buf16(plan, B, fh, fw, 27, 28, dtype=torch.float32)
        self._op(plan, "shape_align.offsets", "align", lambda st: _hip.check(L.m3d_align_offsets(
            0, sel_idx.data_ptr(), sel_prob.data_ptr(), 0.5, P["shape.table"].data_ptr(), None, None, None, 0.0, 1.0, 0.0,
            1.0, om_sa.ptr, om_sa.cs, B, A, HW, 9, 0, st)))
        feats = self._buf16(plan, B, fh, fw, 128, name="feats")
        self._pconv(plan, "shape_align.dcn", P["shape_align"], feats0, feats, 1, 1, act=0, res=feats0, om=om_sa,
                    patch=SHAPE_PATCH)   # anchor-shaped offsets (up to half an anchor): which tiles fit the LDS window is decided per tile
        heads(["bbox_x", "bbox_y"], feats, 0)
        heads(["bbox_x3d", "bbox_y3d"], feats, 4)

        def center_align(p, x, kx, ky, mi, out):
            om = self._buf16(plan, B, fh, fw, 3, 4, dtype=torch.float32)
            self._op(plan, p + ".offsets", "align", lambda st: _hip.check(L.m3d_align_offsets(
                1, sel_idx.data_ptr(), sel_prob.data_ptr(), 0.5, None, box_ptr(kx), box_ptr(ky),
                P["anchor_wh"].data_ptr(), float(means[mi]), float(stds[mi]), float(means[mi + 1]),
                float(stds[mi + 1]), om.ptr, om.cs, B, A, HW, 1, 11 * A * HW, st)))
            self._pconv(plan, p + ".dcn", P[p], x, out, 1, 0, act=0, res=x, om=om)

        f2d = self._buf16(plan, B, fh, fw, 128, name="feats_align2d")
        center_align("center_align2d", feats, 0, 1, 0, f2d)
        f3d = self._buf16(plan, B, fh, fw, 128, name="feats_align3d")
        center_align("center_align3d", feats, 4, 5, 4, f3d)
        # ANAB + the z3d head (reads feats_align3d, writes its own planar rows) next to the six size / orientation heads: the pooling and
        # attention launches leave most of the chip idle (a reduction over pixels, 337 keys) -- as a side branch of the plan they run
        # beside the head launches (Engine._run_plan; M3D_BF16_BRANCH=0: one stream)
        # Measured again with the planar detector (tools/ab_replay.sh, one lease): 10.35-10.40 ms with the branch, without it, and with a
        # reordered tail that puts the pooling beside the x / y heads + the 2-D alignment: the head kernel is persistent and owns every
        # register of every CU, so a side branch only fills the wind-down of its launches -- the step is the SUM of its kernels.
        b0 = len(plan.ops)
        gl = self._buf16(plan, B, fh, fw, 128, name="feats_gl")
        self._anab_bf16(plan, f3d, gl)
        heads(["bbox_z3d"], gl, 6)
        b1 = len(plan.ops)
        heads(["bbox_w", "bbox_h"], f2d, 2)
        heads(["bbox_w3d", "bbox_h3d", "bbox_l3d", "bbox_rY3d"], f3d, 7)
        if BRANCH:
            plan.branches.append((b0, b1, len(plan.ops)))

        cls = torch.empty(B, R, NC, device=self.device, dtype=torch.float32)
        prob = torch.empty(B, R, NC, device=self.device, dtype=torch.float32)
        b2 = torch.empty(B, R, 4, device=self.device, dtype=torch.float32)
        b3 = torch.empty(B, R, 7, device=self.device, dtype=torch.float32)
        key = torch.empty(B, R, device=self.device, dtype=torch.int32)
        plan.named.update(cls=cls, prob=prob, bbox_2d=b2, bbox_3d=b3, score_bits=key)
        # destination of the four bundled outputs: the plan's own buffers, or -- for one forward -- fresh tensors handed in by
        # Engine.forward(fresh=True) (RPN.forward returns fresh tensors like the reference: written in place, no copies)
        dst = plan.named["out_dst"] = [cls, prob, b2, b3]
        self._op(plan, "bundle_outputs", "bundle", lambda st: _hip.check(L.m3d_bundle_outputs(
            cls_pl.data_ptr(), box_pl.data_ptr(), dst[0].data_ptr(), dst[1].data_ptr(), dst[2].data_ptr(), dst[3].data_ptr(),
            key.data_ptr(), B, A, HW, st)), nbytes=B * R * (NC + 11 + 2 * NC + 4 + 7 + 1) * 4)
        return plan

    def _anab_bf16(self, plan, x, out):
        """ANAB (attention.py:183-216) + the BN / LeakyReLU that follows it: Q as bf16, K|V|S in fp32 for the gated pyramid
        pooling (fp32 sums over up to 7680 pixels), pooled keys / values converted to bf16 operands, logits in fp32, softmax
        -> bf16 probabilities, P.V with the residual + BN + LeakyReLU epilogue."""
        L, P = self.L, self.P
        B, fh, fw = x.n, x.h, x.w
        HW = fh * fw
        if HW % 128:
            raise RuntimeError("bf16 ANAB: H*W of the feature map (%dx%d) must be a multiple of 128 (per-image GEMM operands)" % (fh, fw))
        nested = fh % 16 == 0 and fw % 16 == 0 and PSP_SIZES == (1, 4, 8, 16)
        ck, cv, ns, ck_pad = self.ck, self.cv, self.ns, self.ck_pad
        n_bins = sum(s * s for s in PSP_SIZES)
        keys_pad = _rup(n_bins, 64)
        q = self._buf16(plan, B, fh, fw, ck_pad, zero=True)       # channels [ck, ck_pad) are never written: they must be 0, not NaN
        ckvs = ck + cv + ns
        kv16 = KV_BF16 and nested and (ck + cv) % 8 == 0
        qkvs1 = (kv16 and HEADS2 and x.c == 128 and ck_pad % 8 == 0 and ns <= 8 and ck_pad + ck + cv + 8 <= 512)
        if not qkvs1:
            self._pconv(plan, "anab.q", P["anab.q"], x, q, 1, 0, affine=False)
        if qkvs1:
            # query | key | value | gates as ONE launch over the shared input (csrc/bf16_head_mlp2.hip: bf16_qkvs_kernel)
            kvb = self._buf16(plan, B, fh, fw, ck + cv)
            sg = self._buf16(plan, B, fh, fw, ns, _rup(ns, 4), dtype=torch.float32)
            if "anab.qkvs_frag" not in P:
                a0 = "bbox_z3d_gl.0"
                wq, wk, wv, ws_ = (self.sd[a0 + n].detach().cpu().float().reshape(-1, 128) for n in
                                   (".query_conv.weight", ".key_conv.weight", ".value_conv.weight", ".spatial_conv.weight"))
                stack = torch.zeros(512, 128)
                stack[:ck] = wq
                stack[ck_pad:ck_pad + ck] = wk
                stack[ck_pad + ck:ck_pad + ck + cv] = wv
                stack[ck_pad + ck + cv:ck_pad + ck + cv + ns] = ws_
                P["anab.qkvs_frag"] = torch.cat([_head2_frag(stack[256 * i:256 * (i + 1)], BF16) for i in range(2)], 0).contiguous().to(self.device)
            d = QkvsBf16Desc()
            d.inp, d.in_cs, d.M, d.wf = x.ptr, x.cs, B * HW, P["anab.qkvs_frag"].data_ptr()
            d.q, d.q_cs, d.q_rows = q.ptr, q.cs, ck_pad
            d.kv, d.kv_cs, d.kv_rows = kvb.ptr, kvb.cs, ck + cv
            d.s, d.s_cs, d.s_rows = sg.ptr, sg.cs, ns
            ref = ctypes.byref(d)
            plan.keep.append(P["anab.qkvs_frag"])
            plan.ops.append(("anab.qkvs", "bf16_qkvs", 2.0 * B * HW * 128 * (ck + ck + cv + ns),
                             lambda st: _hip.check(L.m3d_anab_qkvs_bf16_forward(ref, st)),
                             OpCost(B * HW * (128 * 2 + (ck_pad + ck + cv) * 2 + 4 * sg.cs) + 512 * 128 * 2)))
        elif kv16:
            # K|V as bf16 NHWC (half the bytes written here and read by the pooling: 590 -> 300 MB at bs = 64), gates in fp32
            kvb = self._buf16(plan, B, fh, fw, ck + cv)
            sg = self._buf16(plan, B, fh, fw, ns, _rup(ns, 4), dtype=torch.float32)
            self._pconv(plan, "anab.kv", P["anab.kv"], x, kvb, 1, 0, affine=False)
            self._pconv(plan, "anab.s", P["anab.s"], x, sg, 1, 0, sigmoid_from=0, out_mode=1, affine=False)
        else:
            kvs = self._buf16(plan, B, fh, fw, ckvs, _rup(ckvs, 4), dtype=torch.float32)
            self._pconv(plan, "anab.kvs", P["anab.kvs"], x, kvs, 1, 0, sigmoid_from=ck + cv, out_mode=1, affine=False)
        khat = torch.zeros(B * keys_pad * ck_pad, device=self.device, dtype=torch.float32)
        vhatT = torch.zeros(B * cv * keys_pad, device=self.device, dtype=torch.float32)
        khat16 = torch.zeros(B * keys_pad * ck_pad, device=self.device, dtype=BF16)
        vhat16 = torch.zeros(B * cv * keys_pad, device=self.device, dtype=BF16)
        plan.keep += [khat, vhatT, khat16, vhat16]
        plan.named["anab.khat"], plan.named["anab.vhatT"] = khat, vhatT
        if kv16:
            scratch = torch.empty(L.m3d_anab_pool_nested_scratch_bytes(B, ck + cv) // 4, device=self.device, dtype=torch.float32)
            plan.keep.append(scratch)
            # (+ the bf16 twins of khat / vhatT the attention kernel reads: no separate conversion launches)
            self._op(plan, "anab.pool_nested", "anab_pool", lambda st: _hip.check(L.m3d_anab_pool_nested_bf16_ex(
                kvb.ptr, kvb.cs, sg.ptr, sg.cs, B, fh, fw, ck, cv, scratch.data_ptr(), khat.data_ptr(), keys_pad, ck_pad,
                vhatT.data_ptr(), 0, khat16.data_ptr(), vhat16.data_ptr(), st)))
        elif nested:      # the windows of the four scales nest (48x160 map): one pass over the features
            kv_ptr, s_ptr = kvs.ptr, kvs.ptr + 4 * (ck + cv)
            scratch = torch.empty(L.m3d_anab_pool_nested_scratch_bytes(B, ck + cv) // 4, device=self.device, dtype=torch.float32)
            plan.keep.append(scratch)
            self._op(plan, "anab.pool_nested", "anab_pool", lambda st: _hip.check(L.m3d_anab_pool_nested(
                kv_ptr, kvs.cs, s_ptr, kvs.cs, B, fh, fw, ck, cv, scratch.data_ptr(), khat.data_ptr(), keys_pad, ck_pad,
                vhatT.data_ptr(), 0, st)))
        else:
            kv_ptr, s_ptr = kvs.ptr, kvs.ptr + 4 * (ck + cv)
            items, bin_scale, bin_slots, bin_inv = self._anab_items(fh, fw)
            max_slots = int(bin_slots.max())
            d_items, d_bscale = torch.from_numpy(items).to(self.device), torch.from_numpy(bin_scale).to(self.device)
            d_bslots, d_binv = torch.from_numpy(bin_slots).to(self.device), torch.from_numpy(bin_inv).to(self.device)
            partial = torch.empty(B * n_bins * max_slots * (ck + cv), device=self.device, dtype=torch.float32)
            plan.keep += [d_items, d_bscale, d_bslots, d_binv, partial]
            self._op(plan, "anab.pool_partial", "anab_pool", lambda st: _hip.check(L.m3d_anab_pool_partial(
                kv_ptr, kvs.cs, s_ptr, kvs.cs, d_items.data_ptr(), items.shape[0], d_bscale.data_ptr(), n_bins,
                partial.data_ptr(), max_slots, B, fh, fw, ck + cv, st)))
            self._op(plan, "anab.pool_finish", "anab_pool", lambda st: _hip.check(L.m3d_anab_pool_finish(
                partial.data_ptr(), d_bslots.data_ptr(), d_binv.data_ptr(), n_bins, max_slots, ck, cv, khat.data_ptr(),
                keys_pad, ck_pad, vhatT.data_ptr(), B, 0, st)))
        if not kv16:
            self._op(plan, "anab.khat_bf16", "convert", lambda st: _hip.check(L.m3d_f32_to_bf16(
                khat.data_ptr(), khat16.data_ptr(), khat.numel(), st)))
            self._op(plan, "anab.vhat_bf16", "convert", lambda st: _hip.check(L.m3d_f32_to_bf16(
                vhatT.data_ptr(), vhat16.data_ptr(), vhatT.numel(), st)))
        if FUSED_ANAB and ck_pad == 192 and cv == 128:
            # logits + softmax + P.V in one launch: the fp32 logits / bf16 probabilities (1.1 GB at bs = 64) never reach HBM
            sc, sh = P["anab.bn.scale"], P["anab.bn.shift"]
            plan.keep += [sc, sh]
            self._op(plan, "anab.attend", "bf16_anab", lambda st: _hip.check(L.m3d_anab_attend_bf16(
                q.ptr, q.cs, khat16.data_ptr(), vhat16.data_ptr(), B, HW, ck_pad, n_bins, keys_pad, cv, x.ptr, x.cs,
                sc.data_ptr(), sh.data_ptr(), 1, out.ptr, out.cs, st)))
            plan.ops[-1] = plan.ops[-1][:2] + (2.0 * B * HW * n_bins * (ck + cv),) + plan.ops[-1][3:]
            return
        logits = self._buf16(plan, B, fh, fw, n_bins, keys_pad, dtype=torch.float32)
        self._conv16(plan, "anab.logits", q, logits, wgt=khat16, kpad=ck_pad, cout=n_bins, cout_pad=keys_pad, out_mode=1,
                     wgt_img_stride=keys_pad * ck_pad, cin=ck_pad, flops_cin=ck)
        pm = self._buf16(plan, B, fh, fw, keys_pad)
        self._op(plan, "anab.softmax", "softmax_bf16", lambda st: _hip.check(L.m3d_softmax_rows_bf16(
            logits.ptr, B * HW, n_bins, keys_pad, pm.ptr, keys_pad, st)))
        self._conv16(plan, "anab.pv", pm, out, wgt=vhat16, kpad=keys_pad, cout=cv, cout_pad=_rup(cv, 32), scale=P["anab.bn.scale"],
                     shift=P["anab.bn.shift"], act=1, res=x, res_mode=1, wgt_img_stride=cv * keys_pad, cin=keys_pad,
                     flops_cin=n_bins)
