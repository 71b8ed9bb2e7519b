"""Inference engine: the whole RPN.forward of M3DSSD as a fixed sequence of HIP kernel launches.

Design (MI355X-first):
  * activations live in HBM as NHWC fp32; channel concatenations (DLA ``Root`` inputs) are never
    materialised -- producers write straight into channel slices of one buffer (``View.slice``);
  * every conv / DCN is one ``m3d_conv2d_forward`` launch with BatchNorm(eval) + bias folded into a
    per-channel affine, residual add and LeakyReLU in the epilogue;
  * parameters are folded / packed once (``Engine.__init__``), buffers and launch descriptors are
    built once per input shape (``_Plan``) and replayed; PyTorch only owns device memory + streams.

Reference graph being executed: model/M3d_inference_align.py:215-313 (RPN.forward),
model/pose_dla_dcn.py:391-397,314-327,546-578,687-696 (DLA-34, Tree, IDAUp, DLAUp, DLASeg),
model/DCNv2/dcn_v2.py:64-70 (DCN), model/module/feturealign_mgpu.py:48-99,153-208,
model/module/attention.py:183-216.
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _hip
from ._hip import ConvDesc, MlpDesc

BN_EPS = 1e-5
SELECT_KEYS = os.environ.get("M3D_SELECT_KEYS", "1") != "0"     # anchor_select also writes the detection stage's sort keys
PSP_SIZES = (1, 4, 8, 16)


def _rup(a, b):
    return (a + b - 1) // b * b


class View:
    """NHWC view (possibly a channel slice) of a device buffer."""
    __slots__ = ("t", "ptr", "n", "h", "w", "c", "cs", "keep")

    def __init__(self, t, n, h, w, c, cs=None, ptr=None):
        self.t, self.n, self.h, self.w, self.c = t, n, h, w, c
        self.cs = c if cs is None else cs
        self.ptr = t.data_ptr() if ptr is None else ptr

    def slice(self, c0, c):
        assert c0 % 4 == 0 and c0 + c <= self.cs
        return View(self.t, self.n, self.h, self.w, c, self.cs, self.ptr + 4 * c0)

    def torch_nchw(self):
        """Debug helper: materialise as an NCHW torch tensor (copies)."""
        full = self.t.view(self.n, self.h, self.w, self.cs)
        off = (self.ptr - self.t.data_ptr()) // 4
        return full[..., off:off + self.c].permute(0, 3, 1, 2).contiguous()


class _Stream:
    @staticmethod
    def current(device=None):
        return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class PackedConv:
    """Packed weights + folded affine of one conv (or DCN main conv)."""

    def __init__(self, eng, weight, bias=None, bn=None, cout_pad_to=32, cin_pad=None):
        w = weight.detach().to(eng.device, torch.float32).contiguous()
        self.cout, self.cin, self.kh, self.kw = w.shape
        self.cin_pad = self.cin if cin_pad is None else cin_pad
        self.cout_pad = _rup(self.cout, cout_pad_to)
        self.wp = torch.empty(self.cout_pad * self.kh * self.kw * self.cin_pad, device=eng.device, dtype=torch.float32)
        _hip.check(eng.L.m3d_pack_conv_weight(w.data_ptr(), self.wp.data_ptr(), self.cout, self.cout_pad, self.cin,
                                              self.cin_pad, self.kh, self.kw, _Stream.current()))
        scale = torch.ones(self.cout, device=eng.device, dtype=torch.float32)
        shift = torch.zeros(self.cout, device=eng.device, dtype=torch.float32)
        if bias is not None:
            shift = bias.detach().to(eng.device, torch.float32).clone()
        if bn is not None:
            g, b, m, v = (t.detach().to(eng.device, torch.float32) for t in bn)
            s = g / torch.sqrt(v + BN_EPS)
            shift = (shift - m) * s + b
            scale = s
        self.scale, self.shift = scale.contiguous(), shift.contiguous()
        self.has_affine = (bias is not None) or (bn is not None)
        self._eng_device = eng.device
        self._frag = None
        self.wino = None
        self._w_cpu = None
        if USE_WINO and self.kh == 3 and self.kw == 3 and self.cin_pad == self.cin and self.cin % 16 == 0 and self.cin >= 32:
            self.wino = pack_wino(w, self.cout_pad, eng.device)
            if USE_WINO44 and self.cout_pad % 64 == 0:
                self._w_cpu = w.detach().cpu()            # F(4x4,3x3) weights are packed on first use (wino44())
        self._wino44 = None

    def wino44(self):
        """U = G g G^T of Winograd F(4x4,3x3) in the fragment order of m3d_wino44_conv3x3_forward, or None."""
        if self._wino44 is None and self._w_cpu is not None:
            self._wino44 = pack_wino44(self._w_cpu, self.cout_pad, self._eng_device)
        return self._wino44


    def frag(self):
        """The packed [Cout_pad, kh*kw*Cin_pad] matrix in MFMA-fragment order (m3d_conv_wave_forward), built on demand."""
        if self._frag is None:
            k = self.kh * self.kw * self.cin_pad
            self._frag = pack_frag(self.wp.view(self.cout_pad, k), self.cout_pad, self._eng_device)
        return self._frag


def pack_frag(weight2d, rows_pad, device):
    """[R, K] -> MFMA-fragment order for m3d_head_mlp_forward: [R/32][K/8][h=2][r=32][t=4] (rows zero-padded)."""
    w = weight2d.detach().to(device, torch.float32)
    r, k = w.shape
    assert k % 8 == 0 and rows_pad % 32 == 0 and rows_pad >= r
    if rows_pad != r:
        w = torch.cat([w, w.new_zeros(rows_pad - r, k)], 0)
    return w.view(rows_pad // 32, 32, k // 8, 2, 4).permute(0, 2, 3, 1, 4).contiguous().view(-1)


_WINO_G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)


def pack_wino(weight, cout_pad, device):
    """[Cout, Cin, 3, 3] -> Winograd F(2x2,3x3) weights U = G g G^T (computed in fp64, rounded once to fp32) in
    MFMA-fragment order [16 xi][Cout_pad/32][Cin/8][h=2][r=32][t=4] for m3d_wino_conv3x3_forward."""
    g = weight.detach().to("cpu", torch.float64)
    co, ci = g.shape[0], g.shape[1]
    assert g.shape[2:] == (3, 3) and ci % 8 == 0 and cout_pad % 32 == 0 and cout_pad >= co
    u = torch.einsum("ai,ocij,bj->aboc", _WINO_G, g, _WINO_G).reshape(16, co, ci)
    if cout_pad != co:
        u = torch.cat([u, u.new_zeros(16, cout_pad - co, ci)], 1)
    u = u.view(16, cout_pad // 32, 32, ci // 8, 2, 4).permute(0, 1, 3, 4, 2, 5).contiguous()
    return u.to(torch.float32).reshape(-1).to(device)


_WINO44_G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                          [0, 0, 1]], dtype=torch.float64)


def pack_wino44(weight, cout_pad, device):
    """[Cout, Cin, 3, 3] -> Winograd F(4x4,3x3) weights U = G g G^T (6 x 6 per filter, computed in fp64, rounded once to fp32) in
    the B-fragment order of v_mfma_f32_16x16x4_f32 for m3d_wino44_conv3x3_forward: [Cout_pad/32][Cin/16][36 xi][j 2][lane 64][e 4]
    with value U[xi][cin = 16 s + 4 (lane >> 4) + e][cout = 32 cb + 16 j + (lane & 15)]."""
    g = weight.detach().to("cpu", torch.float64)
    co, ci = g.shape[0], g.shape[1]
    assert g.shape[2:] == (3, 3) and ci % 16 == 0 and cout_pad % 32 == 0 and cout_pad >= co
    u = torch.einsum("ai,ocij,bj->aboc", _WINO44_G, g, _WINO44_G).reshape(36, co, ci)
    if cout_pad != co:
        u = torch.cat([u, u.new_zeros(36, cout_pad - co, ci)], 1)
    # [xi][cb][j][n16][s][q][e] -> [cb][s][xi][j][q][n16][e]   (lane = q * 16 + n16)
    u = u.view(36, cout_pad // 32, 2, 16, ci // 16, 4, 4).permute(1, 4, 0, 2, 5, 3, 6).contiguous()
    return u.to(torch.float32).reshape(-1).to(device)


def pack_wino44_c16(weight, device):
    """[16, 16, 3, 3] -> U = G g G^T for m3d_conv3x3_c16_wino (DLA level0): [36 xi][64 lanes = 16 (cin / 4) + cout][cin % 4], fp64 -> fp32."""
    g = weight.detach().to("cpu", torch.float64)
    assert tuple(g.shape) == (16, 16, 3, 3)
    u = torch.einsum("ai,ocij,bj->abco", _WINO44_G, g, _WINO44_G).reshape(36, 4, 4, 16)      # xi, cin / 4, cin % 4, cout
    return u.permute(0, 1, 3, 2).contiguous().to(torch.float32).reshape(-1).to(device)


WINO_MIN_BLOCKS = int(os.environ.get("M3D_WINO_MIN_BLOCKS", "128"))
FUSED_ANAB = os.environ.get("M3D_FUSED_ANAB", "1") != "0"       # logits + softmax + P.V of ANAB in one launch (csrc/anab_attend.hip)
USE_ANAB_WAVE = os.environ.get("M3D_ANAB_WAVE", "1") != "0"
USE_ANAB_NESTED = os.environ.get("M3D_ANAB_NESTED", "1") != "0"
USE_DCN_WAVE = os.environ.get("M3D_DCN_WAVE", "1") != "0"
USE_CONV_WAVE = os.environ.get("M3D_CONV_WAVE", "1") != "0"
USE_WINO = os.environ.get("M3D_WINO", "1") != "0"
USE_WINO44 = os.environ.get("M3D_WINO44", "1") != "0"
# F(4x4,3x3) workgroups (16 tiles x 128 or 64 channels, one per CU, 256 CUs): 128-channel workgroups need >= 200 of them; the
# 64-channel form (half the MFMAs per transformed input) pays from ~2 rounds on: in the network, bs 8, level2 (64 -> 64 @ 96x320,
# 960 workgroups) gains 10 %, level4 (256 -> 256 @ 24x80, 240 workgroups) loses 5 % against the F(2x2,3x3) wave kernel
WINO44_TOUCH = os.environ.get("M3D_WINO44_TOUCH", "1") != "0"
WINO44_TOUCH_SPAN = int(os.environ.get("M3D_WINO44_TOUCH_SPAN", "2"))   # launches between two F(4x4) layers that a folded touch bridges
USE_WINO44_SPLITK = os.environ.get("M3D_WINO44_SPLITK", "1") != "0"
W44_SPLIT_NB = 2 if os.environ.get("M3D_W44_SPLIT_NB", "1") == "2" else 1    # form of the split-K launches (csrc/wino44_conv.hip)
W44_SPLIT_KPAIR = os.environ.get("M3D_W44_SPLIT_NB", "1") == "3"             # ... K-pair workgroups inside every slice
# Round 4: the 64-channel form runs TWO workgroups per CU (wino44_kernel<1, 2>) and beats the 128-channel form on every layer
# (128 -> 128 @ 48x160: 0.056 vs 0.065 ms, 128 -> 256: 0.114 vs 0.130); the 128-channel form is kept for experiments
# (M3D_WINO44_MIN_WGS=200 restores the round-3 choice).
WINO44_MIN_WGS = int(os.environ.get("M3D_WINO44_MIN_WGS", "1000000"))
WINO44_MIN_WGS_NB1 = int(os.environ.get("M3D_WINO44_MIN_WGS_NB1", "200"))


class OpCost:
    """Algorithmic HBM bytes of a helper launch (the `desc` slot of a plan op that has no conv / head descriptor)."""

    def __init__(self, hbm_bytes):
        self.hbm_bytes = int(hbm_bytes)


def lo_hi_contains(start, end, n, br):
    """Is the side branch (b0, b1, join) wholly inside ops[start:end]?  (A partial range runs its ops in line.)"""
    lo, hi = start, (n if end is None else end)
    return lo <= br[0] and br[2] <= hi


class _Plan:
    """Buffers + launch list for one input shape."""

    def __init__(self):
        self.ops = []          # (name, callable)
        self.keep = []         # tensors kept alive
        self.named = {}        # name -> View / tensor (taps for tests)
        self.w44_prev = None   # (index into ops, touch record) of the latest F(4x4) launch: it warms the cache for the next one
        self.branches = []     # (b0, b1, join): ops [b0, b1) on a side stream beside ops [b1, join) (Engine._run_plan)


class Engine:
    def __init__(self, state_dict, conf, device=None, backbone_only=False):
        self.L = _hip.lib()
        self.backbone_only = backbone_only
        self.device = torch.device(device if device is not None else conf.device)
        if self.device.type != "cuda":
            raise NotImplementedError("the M3DSSD HIP engine runs on a ROCm device only (got %s)" % self.device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.conf = conf
        sd = {k[7:] if k.startswith("module.") else k: v for k, v in state_dict.items()}
        self.sd = sd
        self.A = int(np.asarray(conf.anchors).shape[0]) if not backbone_only else 0
        self.NC = len(conf.lbls) + 1 if not backbone_only else 0
        self.stride = int(conf.feat_stride)
        with torch.cuda.device(self.device):
            self._pack(sd)
        self.plans = {}
        self.profile = None    # set to a list to collect (name, kind, flops, ms) per op (HIP events on the launch stream)
        self.profile_kinds = None   # optional set: only ops of these kinds are bracketed with events
        self._pending = []

    # ------------------------------------------------------------------ parameters
    def _bn(self, p):
        sd = self.sd
        return (sd[p + ".weight"], sd[p + ".bias"], sd[p + ".running_mean"], sd[p + ".running_var"])

    def _pc(self, conv, bn=None, **kw):
        sd = self.sd
        return PackedConv(self, sd[conv + ".weight"], sd.get(conv + ".bias"), self._bn(bn) if bn else None, **kw)

    def _pack(self, sd):
        dev = self.device
        P = {}
        b = "base.base"
        # stem 7x7: [16,3,7,7] -> [(i*7+j)*3+c][16]
        w = sd[b + ".base_layer.0.weight"].detach().to(dev, torch.float32)
        P["stem.w"] = w.permute(2, 3, 1, 0).contiguous()
        g, be, m, v = (t.detach().to(dev, torch.float32) for t in self._bn(b + ".base_layer.1"))
        s = g / torch.sqrt(v + BN_EPS)
        P["stem.scale"], P["stem.shift"] = s.contiguous(), (be - m * s).contiguous()
        P["level0"] = self._pc(b + ".level0.0", b + ".level0.1")
        # level0 also as a direct VALU conv: weights [(i*3+j)*16 + cin][cout]
        w0 = sd[b + ".level0.0.weight"].detach().to(dev, torch.float32)
        P["level0.direct"] = w0.permute(2, 3, 1, 0).contiguous() if tuple(w0.shape) == (16, 16, 3, 3) else None
        P["level0.wino44"] = pack_wino44_c16(w0, self.device) if (USE_WINO44 and tuple(w0.shape) == (16, 16, 3, 3)) else None
        P["level1"] = self._pc(b + ".level1.0", b + ".level1.1")

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
            P[p + ".dcn"] = PackedConv(self, sd[p + ".conv.weight"], sd[p + ".conv.bias"], self._bn(p + ".actf.0"),
                                       cout_pad_to=64)

        def ida(p, n):
            for i in range(1, n):
                deform("%s.proj_%d" % (p, i))
                deform("%s.node_%d" % (p, i))
                up = sd["%s.up_%d.weight" % (p, i)].detach().to(dev, torch.float32)     # [C,1,4,4]
                P["%s.up_%d" % (p, i)] = up[:, 0].permute(1, 2, 0).contiguous()          # [4][4][C]

        ida("base.dla_up.ida_0", 2)
        ida("base.dla_up.ida_1", 3)
        ida("base.ida_up", 2)

        self.P = P
        if not self.backbone_only:
            self._pack_heads(sd, P)
        torch.cuda.synchronize(self.device)

    def _pack_anab(self, P, wq, wk, wv, ws):
        """One fused 1x1 conv producing [Q (168 -> 192 pad) | K 168 | V 128 | S 4], sigmoid on S."""
        self.ck, self.cv, self.ns = wq.shape[0], wv.shape[0], ws.shape[0]
        self.ck_pad = _rup(self.ck, 32)
        cin = wq.shape[1]
        zpad = torch.zeros(self.ck_pad - self.ck, cin, 1, 1)
        wall = torch.cat([wq.detach().cpu().float(), zpad, wk.detach().cpu().float(), wv.detach().cpu().float(),
                          ws.detach().cpu().float()], 0)
        P["anab.qkvs"] = PackedConv(self, wall, None, None)
        self.anab_off = dict(q=0, k=self.ck_pad, v=self.ck_pad + self.ck, s=self.ck_pad + self.ck + self.cv)
        self.anab_ctot = wall.shape[0]

    def _pack_heads(self, sd, P):
        dev = self.device

        def head(p):
            P[p + ".0"] = self._pc(p + ".0", p + ".1")
            P[p + ".3"] = self._pc(p + ".3", p + ".4")
            P[p + ".6"] = self._pc(p + ".6", cout_pad_to=256 if p == "cls" else 64)   # fused-MLP output tile
            for li, rows in ((".0", 256), (".3", 256), (".6", P[p + ".6"].cout_pad)):
                wt = sd[p + li + ".weight"]
                if wt.shape[2] == 1:                                                    # 1x1 layers only
                    P[p + li + ".frag"] = pack_frag(wt.reshape(wt.shape[0], wt.shape[1]), rows, dev)

        self.box_heads = ["bbox_x", "bbox_y", "bbox_w", "bbox_h", "bbox_x3d", "bbox_y3d", "bbox_z3d", "bbox_w3d",
                          "bbox_h3d", "bbox_l3d", "bbox_rY3d"]
        for h in ["cls"] + self.box_heads:
            head(h)
        for p in ("shape_align", "center_align2d", "center_align3d"):
            P[p] = PackedConv(self, sd[p + ".align.weight"], sd[p + ".align.bias"], None, cout_pad_to=64)

        a = "bbox_z3d_gl.0"
        self._pack_anab(P, sd[a + ".query_conv.weight"], sd[a + ".key_conv.weight"], sd[a + ".value_conv.weight"],
                        sd[a + ".spatial_conv.weight"])
        g, be, m, v = (t.detach().to(dev, torch.float32) for t in self._bn("bbox_z3d_gl.1"))
        s = g / torch.sqrt(v + BN_EPS)
        P["anab.bn.scale"], P["anab.bn.shift"] = s.contiguous(), (be - m * s).contiguous()

        # constants of the align stages
        anchors = torch.as_tensor(np.asarray(self.conf.anchors), dtype=torch.float32)
        aw = (anchors[:, 2] - anchors[:, 0])
        ah = (anchors[:, 3] - anchors[:, 1])
        tab = torch.zeros(self.A, 18, dtype=torch.float32)
        h_step, w_step = ah / self.stride / 3, aw / self.stride / 3        # feturealign_mgpu.py:119-136
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
    def _buf(self, plan, n, h, w, c, cs=None, name=None, zero=False):
        cs = c if cs is None else cs
        t = (torch.zeros if zero else torch.empty)(n * h * w * cs, device=self.device, dtype=torch.float32)
        plan.keep.append(t)
        v = View(t, n, h, w, c, cs)
        if name:
            plan.named[name] = v
        return v

    def _conv(self, plan, name, pc, x, out, stride=1, pad=0, act=0, res=None, res_mode=0, sigmoid_from=-1, om=None,
              planar=None, affine=True, wgt_ptr=None, wgt_img_stride=0, cout=None, cout_pad=None, scale=None, shift=None,
              kh=None, kw=None, cin_true=None, wgt_frag=False):
        d = ConvDesc()
        d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.ptr, x.cs, x.n, x.h, x.w, x.c
        kh = pc.kh if kh is None else kh
        kw = pc.kw if kw is None else kw
        d.wgt = pc.wp.data_ptr() if wgt_ptr is None else wgt_ptr
        d.wgt_img_stride = wgt_img_stride
        d.Cout = pc.cout if cout is None else cout
        d.Cout_pad = pc.cout_pad if cout_pad is None else cout_pad
        d.kh, d.kw, d.stride, d.pad, d.dil = kh, kw, stride, pad, 1
        d.Ho = (x.h + 2 * pad - kh) // stride + 1
        d.Wo = (x.w + 2 * pad - kw) // stride + 1
        if planar is not None:
            t, img_stride, ch_off = planar
            d.out, d.out_nchw, d.out_img_stride = t.data_ptr() + 4 * ch_off * d.Ho * d.Wo, 1, img_stride
        else:
            assert out.h == d.Ho and out.w == d.Wo and out.c >= d.Cout, (name, out.h, out.w, d.Ho, d.Wo)
            d.out, d.out_cs = out.ptr, out.cs
        if scale is not None:
            d.scale, d.shift = scale.data_ptr(), shift.data_ptr()
            plan.keep += [scale, shift]
        elif affine and pc is not None and pc.has_affine:
            d.scale, d.shift = pc.scale.data_ptr(), pc.shift.data_ptr()
        if res is not None:
            d.res, d.res_cs, d.res_mode = res.ptr, res.cs, res_mode
        d.act, d.sigmoid_from = act, sigmoid_from
        if om is not None:
            d.dcn_offmask, d.dcn_om_cs = om.ptr, om.cs
        L = self.L
        ref = ctypes.byref(d)
        # the Winograd kernel works on 256-pixel x 32-channel tiles: a narrow layer (the 27-channel DCN offset/mask
        # convs) on a small map yields < 128 workgroups and goes to the split-K igemm instead
        wino_blocks = -(-(x.n * x.h * x.w) // 256) * (d.Cout_pad // 32)
        strips = -(-(x.n * x.h * x.w) // 256)
        nb = 2 if (d.Cout_pad % 128 == 0 and strips * (d.Cout_pad // 128) >= WINO44_MIN_WGS) else \
            (1 if strips * (d.Cout_pad // 64) >= WINO44_MIN_WGS_NB1 else 0)
        w44ok = (pc is not None and wgt_ptr is None and getattr(pc, "_w_cpu", None) is not None and om is None and planar is None
                 and sigmoid_from < 0 and L.m3d_wino44_applicable(ref) == 1)
        ks44 = 1
        if w44ok and not nb and USE_WINO44_SPLITK:
            # small maps (256 -> 256 @ 24x80, 512 -> 512 @ 12x40 at bs 8): K slices as gridDim.z fill the chip, a second launch
            # adds the partial outputs in slice order
            ssplits, sbytes = ctypes.c_int(), ctypes.c_longlong()
            _hip.check(L.m3d_wino44_splitk_plan(ref, ctypes.byref(ssplits), ctypes.byref(sbytes)))
            if ssplits.value > 1:
                ws = torch.empty(sbytes.value // 4, device=self.device, dtype=torch.float32)
                plan.keep.append(ws)
                d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), sbytes.value
                nb, ks44 = W44_SPLIT_NB, ssplits.value
        if w44ok and nb:
            # Winograd F(4x4,3x3): 4x fewer MFMA FLOPs than the direct convolution (F(2x2,3x3): 2.25x) where the layer fills the
            # chip with 16-tile workgroups (csrc/wino44_conv.hip)
            u44 = pc.wino44()
            plan.keep.append(u44)
            d.wgt = u44.data_ptr()
            flops = 2.0 * x.n * d.Ho * d.Wo * d.Cout * 9 * pc.cin
            # U (2.4-38 MB) is cold in HBM when the layer starts and the kernel's B fragments run only ~1400 cycles ahead of the
            # MFMAs: the PREVIOUS F(4x4) launch touches this layer's U (every thread a few lines, under its own first loads); a
            # layer with no F(4x4) launch shortly before it gets a touch launch of its own
            nxt = {"ptr": None, "bytes": 0}
            if WINO44_TOUCH:
                nbytes = u44.numel() * 4
                if plan.w44_prev is not None and len(plan.ops) - plan.w44_prev[0] <= WINO44_TOUCH_SPAN:
                    plan.w44_prev[1]["ptr"], plan.w44_prev[1]["bytes"] = u44.data_ptr(), nbytes
                else:
                    plan.ops.append((name + ".touch", "touch", 0.0,
                                     lambda st: _hip.check(L.m3d_cache_touch(u44.data_ptr(), nbytes, st)), OpCost(nbytes)))
            plan.w44_prev = (len(plan.ops), nxt)
            kp = ",kpair" if (nb == 1 and ((ks44 == 1 and L.m3d_wino44_kpair(ref) == 1) or (ks44 > 1 and W44_SPLIT_KPAIR))) else ""
            plan.ops.append((name, "wino44<16,%d%s%s>" % (16 * nb, ",splitk%d" % ks44 if ks44 > 1 else "", kp), flops,
                             lambda st: _hip.check(L.m3d_wino44_conv3x3_forward_touch(ref, nb, nxt["ptr"], nxt["bytes"], st)), d))
            return
        if (pc is not None and wgt_ptr is None and getattr(pc, "wino", None) is not None and kh == 3 and kw == 3
                and stride == 1 and pad == 1 and om is None and planar is None and x.h % 2 == 0 and x.w % 2 == 0
                and wino_blocks >= WINO_MIN_BLOCKS):
            # Winograd F(2x2,3x3): 2.25x fewer MFMA FLOPs for the plain 3x3 stride-1 layers
            d.wgt = pc.wino.data_ptr()
            flops = 2.0 * x.n * d.Ho * d.Wo * d.Cout * 9 * pc.cin
            ssplits, sbytes = ctypes.c_int(), ctypes.c_longlong()          # thin layers: wave kernel split along K across waves
            _hip.check(L.m3d_wino_conv3x3_splitk_plan(ref, ctypes.byref(ssplits), ctypes.byref(sbytes)))
            if ssplits.value > 1:
                ws = torch.empty(sbytes.value // 4, device=self.device, dtype=torch.float32)
                plan.keep.append(ws)
                d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), sbytes.value
            if L.m3d_wino_conv3x3_variant(ref) == 1:
                kind = "wino_wave<32,32%s>" % (",splitk%d" % ssplits.value if ssplits.value > 1 else "")
            else:
                kind = "wino_lds<64,32,16>"
            plan.ops.append((name, kind, flops, lambda st: _hip.check(L.m3d_wino_conv3x3_forward(ref, st)), d))
            return
        flops_true = 2.0 * x.n * d.Ho * d.Wo * d.Cout * kh * kw * (
            cin_true if cin_true is not None else (pc.cin if (pc is not None and wgt_ptr is None) else x.c))
        if wgt_frag:
            # explicit fragment-ordered (per-image) weights: the caller has already decided for the wave-granular kernel
            assert wgt_ptr is not None and planar is None and x.cs % 32 == 0 and x.ptr % 128 == 0, name
            if L.m3d_conv_wave_applicable(ref) <= 0:
                raise RuntimeError("%s: fragment-ordered weights but the wave kernel does not apply" % name)
            plan.ops.append((name, "conv_wave", flops_true, lambda st: _hip.check(L.m3d_conv_wave_forward(ref, st)), d))
            return
        if (pc is not None and wgt_ptr is None and planar is None and (USE_DCN_WAVE if om is not None else USE_CONV_WAVE)
                and x.cs % 32 == 0 and x.ptr % 128 == 0 and pc.cin_pad == x.c):
            # wave-granular kernel, no workgroup barriers (csrc/dcn_wave.hip); thin layers are split along K across waves
            wsplits, wbytes = ctypes.c_int(), ctypes.c_longlong()
            _hip.check(L.m3d_conv_wave_splitk_plan(ref, ctypes.byref(wsplits), ctypes.byref(wbytes)))
            ws = None
            if wsplits.value > 1:
                ws = torch.empty(wbytes.value // 4, device=self.device, dtype=torch.float32)
                d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), wbytes.value
            if L.m3d_conv_wave_applicable(ref) > 0:
                frag = pc.frag()
                d.wgt = frag.data_ptr()
                if ws is not None:
                    plan.keep.append(ws)
                kind = "conv_wave%s" % ("<%s>" % ",".join(
                    (["deform"] if om is not None else []) + (["splitk%d" % wsplits.value] if wsplits.value > 1 else []))
                    if (om is not None or wsplits.value > 1) else "")
                plan.ops.append((name, kind, flops_true, lambda st: _hip.check(L.m3d_conv_wave_forward(ref, st)), d))
                return
            d.splitk_ws, d.splitk_ws_bytes = None, 0
        bm, bn, bk, grid = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _hip.check(L.m3d_conv2d_tile(ref, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(bk), ctypes.byref(grid)))
        splits, ws_bytes = ctypes.c_int(), ctypes.c_longlong()
        _hip.check(L.m3d_conv2d_splitk_plan(ref, ctypes.byref(splits), ctypes.byref(ws_bytes)))
        if splits.value > 1:
            ws = torch.empty(ws_bytes.value // 4, device=self.device, dtype=torch.float32)
            plan.keep.append(ws)
            d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws_bytes.value
        kind = "igemm<%d,%d,%d%s%s%s>" % (bm.value, bn.value, bk.value, ",deform" if om is not None else "",
                                          ",planar" if planar is not None else "",
                                          ",splitk%d" % splits.value if splits.value > 1 else "")
        # algorithmic FLOPs of the layer: 2 * pixels * Cout * taps * Cin (true channel counts, no padding)
        if cin_true is None:
            cin_true = pc.cin if (pc is not None and wgt_ptr is None) else x.c
        flops = 2.0 * x.n * d.Ho * d.Wo * d.Cout * kh * kw * cin_true
        plan.ops.append((name, kind, flops, lambda st: _hip.check(L.m3d_conv2d_forward(ref, st)), d))

    def _op(self, plan, name, kind, fn, flops=0.0, nbytes=None):
        """A launch outside the conv / head descriptors: `flops` = arithmetic it executes, `nbytes` = the HBM bytes it has to
        move once (inputs + outputs), so that bench.py can hold it against its own roof (HBM for the element-wise helpers)."""
        plan.ops.append((name, kind, float(flops), fn, OpCost(nbytes) if nbytes is not None else None))

    # ------------------------------------------------------------------ plan construction
    def _build_plan(self, B, H, W):
        L, P = self.L, self.P
        plan = _Plan()
        chs = [16, 32, 64, 128, 256, 512]
        b = "base.base"
        in_ptr = [0]                      # set by forward(): the caller's NCHW tensor is read in place
        plan.named["input_ptr"] = in_ptr
        s0 = self._buf(plan, B, H, W, 16)
        in_u8 = [0, 0, 0]                 # (ptr, h, w) of uint8 BGR frames; set by forward_u8() instead of in_ptr
        plan.named["input_u8"] = in_u8
        mean3 = (ctypes.c_float * 3)(*[float(v) for v in self.conf.image_means])
        stds3 = (ctypes.c_float * 3)(*[float(v) for v in self.conf.image_stds])

        def stem(st):
            if in_u8[0]:                  # test-time input path fused into the stem's loads
                _hip.check(L.m3d_stem_conv7x7_u8(in_u8[0], in_u8[1], in_u8[2], mean3, stds3, P["stem.w"].data_ptr(),
                                                 P["stem.scale"].data_ptr(), P["stem.shift"].data_ptr(), s0.ptr, s0.cs, B, H,
                                                 W, st))
            else:
                _hip.check(L.m3d_stem_conv7x7(in_ptr[0], P["stem.w"].data_ptr(), P["stem.scale"].data_ptr(),
                                              P["stem.shift"].data_ptr(), s0.ptr, s0.cs, B, H, W, st))
        # (bytes: the fp32 NCHW image; with uint8 frames the input is 4x smaller)
        self._op(plan, "stem", "stem", stem, flops=2.0 * B * H * W * 16 * 147, nbytes=B * H * W * (3 + 16) * 4)
        l0 = self._buf(plan, B, H, W, 16, name="level0")
        if P.get("level0.wino44") is not None and H % 4 == 0 and W % 4 == 0 and os.environ.get("M3D_LEVEL0_WINO44", "1") != "0":
            pc0 = P["level0"]       # F(4x4,3x3) on 16x16x4 MFMAs: 144 MFMAs per 256 pixels instead of 576 (csrc/wino44_conv.hip)
            self._op(plan, "level0", "wino44_c16", lambda st: _hip.check(L.m3d_conv3x3_c16_wino(
                s0.ptr, s0.cs, P["level0.wino44"].data_ptr(), pc0.scale.data_ptr(), pc0.shift.data_ptr(), l0.ptr, l0.cs,
                B, H, W, st)), flops=2.0 * B * H * W * 16 * 144, nbytes=B * H * W * 32 * 4)
        elif P["level0.direct"] is not None and os.environ.get("M3D_LEVEL0_IGEMM", "0") != "1":
            pc0 = P["level0"]
            self._op(plan, "level0", "conv3x3_c16", lambda st: _hip.check(L.m3d_conv3x3_c16(
                s0.ptr, s0.cs, P["level0.direct"].data_ptr(), pc0.scale.data_ptr(), pc0.shift.data_ptr(), l0.ptr, l0.cs,
                B, H, W, st)), flops=2.0 * B * H * W * 16 * 144, nbytes=B * H * W * 32 * 4)
        else:
            self._conv(plan, "level0", P["level0"], s0, l0, 1, 1, act=1)
        l1 = self._buf(plan, B, H // 2, W // 2, 32, name="level1")
        self._conv(plan, "level1", P["level1"], l0, l1, 2, 1, act=1)

        def maxpool(name, x, out):
            self._op(plan, name, "maxpool", lambda st: _hip.check(L.m3d_maxpool2x2(
                x.ptr, x.cs, out.ptr, out.cs, x.n, x.h, x.w, x.c, st)), nbytes=x.n * x.h * x.w * x.c * 5)

        def block(p, x, res, out, stride):
            co = P[p + ".conv1"].cout
            t = self._buf(plan, B, out.h, out.w, co)
            self._conv(plan, p + ".conv1", P[p + ".conv1"], x, t, stride, 1, act=1)
            self._conv(plan, p + ".conv2", P[p + ".conv2"], t, out, 1, 1, act=1, res=res)

        def tree1(p, x, co, stride, out, bottom=None):
            """Tree(levels=1, level_root=False): pose_dla_dcn.py:314-323; root input = (x2, x1)."""
            h, w = x.h // stride, x.w // stride
            cat = self._buf(plan, B, h, w, 2 * co)
            x2v, x1v = cat.slice(0, co), cat.slice(co, co)
            if stride == 1:
                bottom = x
            elif bottom is None:
                bottom = self._buf(plan, B, h, w, x.c)
                maxpool(p + ".downsample", x, bottom)
            if (p + ".project") in P:
                res = self._buf(plan, B, h, w, co)
                self._conv(plan, p + ".project", P[p + ".project"], bottom, res, 1, 0, act=0)
            else:
                res = bottom
            block(p + ".tree1", x, res, x1v, stride)
            block(p + ".tree2", x1v, x1v, x2v, 1)
            self._conv(plan, p + ".root", P[p + ".root"], cat, out, 1, 0, act=1)

        # level2: Tree(1, 32 -> 64, stride 2)
        l2 = self._buf(plan, B, H // 4, W // 4, 64, name="level2")
        tree1(b + ".level2", l1, 64, 2, l2)

        def tree2(p, x, co, out):
            """Tree(levels=2, level_root=True): pose_dla_dcn.py:314-327."""
            ci = x.c
            h, w = x.h // 2, x.w // 2
            # tree2's root input: (x2'', x1'', bottom, X1)
            catb = self._buf(plan, B, h, w, 2 * co + ci + co)
            bottom = catb.slice(2 * co, ci)
            X1 = catb.slice(2 * co + ci, co)
            maxpool(p + ".downsample", x, bottom)
            tree1(p + ".tree1", x, co, 2, X1, bottom=bottom)
            # tree2 = Tree(1, co -> co, stride 1) writing x2'', x1'' into catb[0:2co]
            x2v, x1v = catb.slice(0, co), catb.slice(co, co)
            block(p + ".tree2.tree1", X1, X1, x1v, 1)
            block(p + ".tree2.tree2", x1v, x1v, x2v, 1)
            self._conv(plan, p + ".tree2.root", P[p + ".tree2.root"], catb, out, 1, 0, act=1)

        l3 = self._buf(plan, B, H // 8, W // 8, 128, name="level3")
        tree2(b + ".level3", l2, 128, l3)
        l4 = self._buf(plan, B, H // 16, W // 16, 256, name="level4")
        tree2(b + ".level4", l3, 256, l4)
        # level5: Tree(1, 256 -> 512, stride 2, level_root): root input (x2, x1, bottom)
        l5 = self._buf(plan, B, H // 32, W // 32, 512, name="level5")
        h5, w5 = H // 32, W // 32
        cat5 = self._buf(plan, B, h5, w5, 1024 + 256)
        bottom5 = cat5.slice(1024, 256)
        maxpool(b + ".level5.downsample", l4, bottom5)
        res5 = self._buf(plan, B, h5, w5, 512)
        self._conv(plan, b + ".level5.project", P[b + ".level5.project"], bottom5, res5, 1, 0, act=0)
        block(b + ".level5.tree1", l4, res5, cat5.slice(512, 512), 2)
        block(b + ".level5.tree2", cat5.slice(512, 512), cat5.slice(512, 512), cat5.slice(0, 512), 1)
        self._conv(plan, b + ".level5.root", P[b + ".level5.root"], cat5, l5, 1, 0, act=1)

        # ---- DLAUp / IDAUp ------------------------------------------------------------
        def deform(p, x, out):
            om = self._buf(plan, B, x.h, x.w, 27, 28)
            self._conv(plan, p + ".offset_mask", P[p + ".om"], x, om, 1, 1, act=0, sigmoid_from=18)
            self._conv(plan, p + ".dcn", P[p + ".dcn"], x, out, 1, 1, act=1, om=om)
            plan.named[p + ".out"] = out

        def ida_step(p, i, x, skip, co):
            proj = self._buf(plan, B, x.h, x.w, co)
            deform("%s.proj_%d" % (p, i), x, proj)
            summed = self._buf(plan, B, 2 * x.h, 2 * x.w, co)
            upw = P["%s.up_%d" % (p, i)]
            self._op(plan, "%s.up_%d" % (p, i), "upsample", lambda st: _hip.check(L.m3d_upsample2x_add(
                proj.ptr, proj.cs, upw.data_ptr(), skip.ptr, skip.cs, summed.ptr, summed.cs, B, proj.h, proj.w, co, st)),
                flops=2.0 * B * 4 * proj.h * proj.w * co * 4, nbytes=B * proj.h * proj.w * co * 4 * (1 + 4 + 4))
            node = self._buf(plan, B, 2 * x.h, 2 * x.w, co)
            deform("%s.node_%d" % (p, i), summed, node)
            return node

        L5a = ida_step("base.dla_up.ida_0", 1, l5, l4, 256)            # 256 @ H/16
        L4b = ida_step("base.dla_up.ida_1", 1, l4, l3, 128)            # 128 @ H/8
        L5b = ida_step("base.dla_up.ida_1", 2, L5a, L4b, 128)          # 128 @ H/8
        feats0 = ida_step("base.ida_up", 1, L5a, L5b, 128)             # 128 @ H/8
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

        def head_desc(p, x, planar, first):
            t, img_stride, ch_off = planar
            d = MlpDesc()
            d.inp, d.in_cs, d.M, d.Cin = x.ptr, x.cs, B * HW, x.c
            if first is not None:
                d.w1, d.s1, d.t1 = P[p + ".0.frag"].data_ptr(), first.scale.data_ptr(), first.shift.data_ptr()
            mid, last = P[p + ".3"], P[p + ".6"]
            d.w2, d.s2, d.t2 = P[p + ".3.frag"].data_ptr(), mid.scale.data_ptr(), mid.shift.data_ptr()
            d.w3, d.s3, d.t3 = P[p + ".6.frag"].data_ptr(), last.scale.data_ptr(), last.shift.data_ptr()
            d.Cout, d.Cout_pad = last.cout, last.cout_pad
            d.out, d.out_img_stride, d.HW = t.data_ptr() + 4 * ch_off * HW, img_stride, HW
            flops = 2.0 * B * HW * ((x.c * 256 if first is not None else 0) + 256 * 256 + 256 * last.cout)
            return d, flops

        def heads(items):
            """1x1 heads with no mutual dependency as ONE fused-MLP launch (grid.y = head): items = [(p, x, planar)]."""
            arr = (MlpDesc * len(items))()
            flops = 0.0
            for i, (p, x, planar) in enumerate(items):
                arr[i], f = head_desc(p, x, planar, P[p + ".0"])
                flops += f
            n = len(items)
            name = "+".join(p for p, _, _ in items) + ".mlp"
            plan.ops.append((name, "head_mlp<3,%d>" % arr[0].Cout_pad, flops,
                             lambda st: _hip.check(L.m3d_head_mlp_forward_batched(arr, n, st)), arr))

        def head(p, x, planar, k0=1):
            """3-layer head.  A 1x1 head runs as one fused-MLP launch; the cls head runs its 3x3 conv through the
            Winograd kernel and the remaining two 1x1 layers fused."""
            if k0 == 1:
                return heads([(p, x, planar)])
            h1 = self._buf(plan, B, fh, fw, 256)
            self._conv(plan, p + ".0", P[p + ".0"], x, h1, 1, k0 // 2, act=1)
            d, flops = head_desc(p, h1, planar, None)
            ref = ctypes.byref(d)
            plan.ops.append((p + ".mlp", "head_mlp<2,%d>" % d.Cout_pad, flops,
                             lambda st: _hip.check(L.m3d_head_mlp_forward(ref, st)), d))

        head("cls", feats0, (cls_pl, NC * A * HW, 0), 3)
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

        def box_planar(k):
            return (box_pl, 11 * A * HW, k * A)

        def box_ptr(k):
            return box_pl.data_ptr() + 4 * k * A * HW          # image 0; per-image stride handled via [B][11][A][HW]

        # shape_align
        om_sa = self._buf(plan, B, fh, fw, 27, 28)
        self._op(plan, "shape_align.offsets", "align", lambda st: _hip.check(L.m3d_align_offsets(
            0, sel_idx.data_ptr(), sel_prob.data_ptr(), 0.5, P["shape.table"].data_ptr(), None, None, None, 0.0, 1.0, 0.0,
            1.0, om_sa.ptr, om_sa.cs, B, A, HW, 9, 0, st)), nbytes=B * HW * (2 + 28) * 4)
        feats = self._buf(plan, B, fh, fw, 128, name="feats")
        self._conv(plan, "shape_align.dcn", P["shape_align"], feats0, feats, 1, 1, act=0, res=feats0, om=om_sa)
        # heads are grouped by the feature map they read (M3d_inference_align.py:139-176): the four centre heads first,
        # then both centre alignments, then the six size / orientation heads.
        heads([("bbox_x", feats, box_planar(0)), ("bbox_y", feats, box_planar(1)),
               ("bbox_x3d", feats, box_planar(4)), ("bbox_y3d", feats, box_planar(5))])

        def center_align(p, x, kx, ky, mi, out):
            om = self._buf(plan, B, fh, fw, 3, 4)
            self._op(plan, p + ".offsets", "align", lambda st: _hip.check(L.m3d_align_offsets(
                1, sel_idx.data_ptr(), sel_prob.data_ptr(), 0.5, None, box_ptr(kx), box_ptr(ky),
                P["anchor_wh"].data_ptr(), float(means[mi]), float(stds[mi]), float(means[mi + 1]),
                float(stds[mi + 1]), om.ptr, om.cs, B, A, HW, 1, 11 * A * HW, st)), nbytes=B * HW * (2 + 2 + 4) * 4)
            self._conv(plan, p + ".dcn", P[p], x, out, 1, 0, act=0, res=x, om=om)

        f2d = self._buf(plan, B, fh, fw, 128, name="feats_align2d")
        center_align("center_align2d", feats, 0, 1, 0, f2d)
        f3d = self._buf(plan, B, fh, fw, 128, name="feats_align3d")
        center_align("center_align3d", feats, 4, 5, 4, f3d)
        heads([("bbox_w", f2d, box_planar(2)), ("bbox_h", f2d, box_planar(3)),
               ("bbox_w3d", f3d, box_planar(7)), ("bbox_h3d", f3d, box_planar(8)),
               ("bbox_l3d", f3d, box_planar(9)), ("bbox_rY3d", f3d, box_planar(10))])

        # ---- ANAB ---------------------------------------------------------------------
        gl = self._buf(plan, B, fh, fw, 128, name="feats_gl")
        self._anab_ops(plan, f3d, gl, P["anab.bn.scale"], P["anab.bn.shift"], act=1, res_mode=1)
        head("bbox_z3d", gl, box_planar(6))

        # ---- outputs ------------------------------------------------------------------
        cls = torch.empty(B, R, NC, device=self.device, dtype=torch.float32)
        prob = torch.empty(B, R, NC, device=self.device, dtype=torch.float32)
        b2 = torch.empty(B, R, 4, device=self.device, dtype=torch.float32)
        b3 = torch.empty(B, R, 7, device=self.device, dtype=torch.float32)
        key = torch.empty(B, R, device=self.device, dtype=torch.int32)   # monotone score bits (u32)
        plan.named.update(cls=cls, prob=prob, bbox_2d=b2, bbox_3d=b3, score_bits=key)
        assert NC == 4, "bundle kernel is written for 4 classes (bg + 3)"
        # destination of the four bundled outputs: the plan's own buffers, or -- for one forward -- fresh tensors handed in by
        # Engine.forward(fresh=True) (RPN.forward returns fresh tensors like the reference: written in place, no copies)
        dst = plan.named["out_dst"] = [cls, prob, b2, b3]
        self._op(plan, "bundle_outputs", "bundle", lambda st: _hip.check(L.m3d_bundle_outputs(
            cls_pl.data_ptr(), box_pl.data_ptr(), dst[0].data_ptr(), dst[1].data_ptr(), dst[2].data_ptr(), dst[3].data_ptr(),
            key.data_ptr(), B, A, HW, st)), nbytes=B * R * (NC + 11 + 2 * NC + 4 + 7 + 1) * 4)
        plan.feat = (fh, fw)
        return plan


    def _anab_ops(self, plan, x, out, scale, shift, act, res_mode):
        """Append the ANAB launches: fused QKVS 1x1 conv -> weighted pyramid pooling -> logits GEMM (per-image
        pooled keys as weights) -> row softmax -> P.V GEMM with the residual (+ optional BN/LeakyReLU) epilogue."""
        L, P = self.L, self.P
        B, fh, fw = x.n, x.h, x.w
        HW = fh * fw
        ctot, off = self.anab_ctot, self.anab_off
        qkvs = self._buf(plan, B, fh, fw, ctot, _rup(ctot, 32))          # 128-byte pixel rows: the wave kernel's gather map
        self._conv(plan, "anab.qkvs", P["anab.qkvs"], x, qkvs, 1, 0, act=0, sigmoid_from=off["s"], affine=False)
        items, bin_scale, bin_slots, bin_inv = self._anab_items(fh, fw)
        n_bins, max_slots = len(bin_scale), int(bin_slots.max())
        # the two per-image GEMMs go to the wave-granular kernel when they yield enough waves; their weights (pooled keys /
        # values) are then written in MFMA-fragment order by the pooling finish and the key count is padded to 128
        fused = FUSED_ANAB and self.cv == 128 and self.ck in (64, 128, 168) and HW % 128 == 0
        wave_ok = (not fused) and USE_ANAB_WAVE and USE_CONV_WAVE and HW % 32 == 0 and self.ck_pad % 32 == 0 and self.cv % 128 == 0
        nw = B * HW // 32
        wave_logits = wave_ok and nw * (_rup(n_bins, 128) // 128) >= 900
        keys_pad = _rup(n_bins, 128) if wave_logits else _rup(n_bins, 32)
        wave_pv = wave_ok and keys_pad % 32 == 0 and nw * (self.cv // 128) >= 900
        frag = (1 if wave_logits else 0) | (2 if wave_pv else 0)
        d_items = torch.from_numpy(items).to(self.device)
        d_bscale = torch.from_numpy(bin_scale).to(self.device)
        d_bslots = torch.from_numpy(bin_slots).to(self.device)
        d_binv = torch.from_numpy(bin_inv).to(self.device)
        ckv = self.ck + self.cv
        partial = torch.empty(B * n_bins * max_slots * ckv, device=self.device, dtype=torch.float32)
        khat = torch.zeros(B * keys_pad * self.ck_pad, device=self.device, dtype=torch.float32)
        vhatT = torch.zeros(B * self.cv * keys_pad, device=self.device, dtype=torch.float32)
        plan.keep += [d_items, d_bscale, d_bslots, d_binv, partial, khat, vhatT]
        plan.named["anab.khat"], plan.named["anab.vhatT"] = khat, vhatT
        kvv, sv = qkvs.slice(off["k"], ckv), qkvs.slice(off["s"], self.ns)
        if fh % 16 == 0 and fw % 16 == 0 and PSP_SIZES == (1, 4, 8, 16) and USE_ANAB_NESTED:
            # the windows of the four scales nest: one pass over the features (csrc/rpn_kernels.hip)
            scratch = torch.empty(L.m3d_anab_pool_nested_scratch_bytes(B, ckv) // 4, device=self.device, dtype=torch.float32)
            plan.keep.append(scratch)
            self._op(plan, "anab.pool_nested", "anab_pool", lambda st: _hip.check(L.m3d_anab_pool_nested(
                kvv.ptr, kvv.cs, sv.ptr, sv.cs, B, fh, fw, self.ck, self.cv, scratch.data_ptr(), khat.data_ptr(), keys_pad,
                self.ck_pad, vhatT.data_ptr(), frag, st)), nbytes=B * HW * (ckv + self.ns) * 4)
        else:
            self._op(plan, "anab.pool_partial", "anab_pool", lambda st: _hip.check(L.m3d_anab_pool_partial(
                kvv.ptr, kvv.cs, sv.ptr, sv.cs, d_items.data_ptr(), items.shape[0], d_bscale.data_ptr(), n_bins,
                partial.data_ptr(), max_slots, B, fh, fw, ckv, st)))
            self._op(plan, "anab.pool_finish", "anab_pool", lambda st: _hip.check(L.m3d_anab_pool_finish(
                partial.data_ptr(), d_bslots.data_ptr(), d_binv.data_ptr(), n_bins, max_slots, self.ck, self.cv,
                khat.data_ptr(), keys_pad, self.ck_pad, vhatT.data_ptr(), B, frag, st)))
        qv = qkvs.slice(off["q"], self.ck_pad)
        if fused:
            # logits + softmax + P.V in one launch: the fp32 logits (7680 x 352 per image) never reach HBM
            rp, rcs = (x.ptr, x.cs) if x is not None else (None, 0)
            sp = scale.data_ptr() if scale is not None else None
            hp = shift.data_ptr() if shift is not None else None
            self._op(plan, "anab.attend", "anab_attend", lambda st: _hip.check(L.m3d_anab_attend_f32(
                qv.ptr, qv.cs, khat.data_ptr(), self.ck_pad, vhatT.data_ptr(), B, HW, self.ck, n_bins, keys_pad, self.cv, rp, rcs,
                res_mode, sp, hp, 1 if act else 0, out.ptr, out.cs, st)), flops=2.0 * B * HW * n_bins * (self.ck + self.cv),
                nbytes=B * HW * (self.ck + 3 * self.cv) * 4 + B * keys_pad * (self.ck + self.cv) * 4)
            return
        logits = self._buf(plan, B, fh, fw, keys_pad)
        self._conv(plan, "anab.logits", None, qv, logits, 1, 0, act=0, affine=False, wgt_ptr=khat.data_ptr(),
                   wgt_img_stride=keys_pad * self.ck_pad, cout=n_bins, cout_pad=keys_pad, kh=1, kw=1, cin_true=self.ck,
                   wgt_frag=wave_logits)
        self._op(plan, "anab.softmax", "softmax", lambda st: _hip.check(L.m3d_softmax_rows(
            logits.ptr, B * HW, n_bins, keys_pad, st)), nbytes=B * HW * n_bins * 8)
        self._conv(plan, "anab.pv", None, logits, out, 1, 0, act=act, res=x, res_mode=res_mode,
                   wgt_ptr=vhatT.data_ptr(), wgt_img_stride=self.cv * keys_pad, cout=self.cv,
                   cout_pad=_rup(self.cv, 32), kh=1, kw=1, scale=scale, shift=shift, cin_true=n_bins, wgt_frag=wave_pv)

    @classmethod
    def anab_standalone(cls, mod, x):
        """ANAB.forward for a stand-alone module: x is an NHWC View; returns the output View."""
        eng = cls.__new__(cls)
        eng.L, eng.device, eng.profile, eng.P = _hip.lib(), x.t.device, None, {}
        eng._pack_anab(eng.P, mod.query_conv.weight, mod.key_conv.weight, mod.value_conv.weight,
                       mod.spatial_conv.weight)
        plan = _Plan()
        out = eng._buf(plan, x.n, x.h, x.w, x.c)
        eng._anab_ops(plan, x, out, None, None, act=0, res_mode=0)
        eng.run_plan(plan)
        out.keep = plan          # keep the intermediate buffers alive until the caller has copied out
        return out

    @staticmethod
    def _anab_items(H, W):
        """Work items of the pyramid pooling: AdaptiveAvgPool2d windows (start = floor(i*H/s),
        end = ceil((i+1)*H/s)) split into chunks of ceil(H/16) rows x ceil(W/4) columns, so that the
        whole-map "size 1" bin is spread over 64 workgroups instead of serialising on one."""
        rchunk, cchunk = max(1, math.ceil(H / 16)), max(1, math.ceil(W / 4))
        items, bin_scale, bin_slots, bin_inv = [], [], [], []
        b = 0
        for si, s in enumerate(PSP_SIZES):
            for i in range(s):
                h0, h1 = (i * H) // s, -((-(i + 1) * H) // s)
                for j in range(s):
                    w0, w1 = (j * W) // s, -((-(j + 1) * W) // s)
                    slot = 0
                    for r0 in range(h0, h1, rchunk):
                        for c0 in range(w0, w1, cchunk):
                            items.append((b, r0, min(r0 + rchunk, h1), c0, min(c0 + cchunk, w1), slot))
                            slot += 1
                    bin_scale.append(si)
                    bin_slots.append(slot)
                    bin_inv.append(1.0 / ((h1 - h0) * (w1 - w0)))
                    b += 1
        return (np.asarray(items, dtype=np.int32), np.asarray(bin_scale, dtype=np.int32),
                np.asarray(bin_slots, dtype=np.int32), np.asarray(bin_inv, dtype=np.float32))

    # ------------------------------------------------------------------ execution
    def plan_for(self, B, H, W):
        key = (B, H, W)
        if key not in self.plans:
            if H % 32 or W % 32:
                raise RuntimeError("input H and W must be multiples of 32 (got %dx%d)" % (H, W))
            with torch.cuda.device(self.device):
                self.plans[key] = self._build_plan(B, H, W)
        return self.plans[key]

    def _run_to(self, plan, fresh):
        """Run the plan; fresh=True: the four bundled outputs are written into newly allocated tensors (returned), the plan's own
        output buffers are left as they are; else into the plan-owned buffers (returned as views)."""
        n = plan.named
        owned = (n["cls"], n["prob"], n["bbox_2d"], n["bbox_3d"])
        if not fresh:
            self.run_plan(plan)
            return owned
        dst = n["out_dst"]
        outs = [torch.empty_like(t) for t in owned]
        dst[:] = outs
        try:
            self.run_plan(plan)
        finally:
            dst[:] = owned
        return tuple(outs)

    def forward(self, x, fresh=False):
        """x: float32 [B,3,H,W] on the engine's device -> (cls, prob, bbox_2d, bbox_3d) device tensors: views of plan-owned
        buffers, overwritten by the next call with the same shape, or (fresh=True) newly allocated tensors."""
        if not x.is_cuda:
            raise NotImplementedError("M3DSSD HIP engine: input must be a ROCm device tensor")
        if x.device != self.device:
            raise RuntimeError("M3DSSD HIP engine on %s got an input on %s" % (self.device, x.device))
        B, _, H, W = x.shape
        plan = self.plan_for(B, H, W)
        x = x.contiguous()
        plan.named["input_ptr"][0] = x.data_ptr()
        plan.named["input_u8"][0] = 0
        return self._run_to(plan, fresh)

    def forward_u8(self, frames, size=None, fresh=False):
        """frames: uint8 [B, h, w, 3] BGR device tensor (what cv2.imread returns, batched) -> the same outputs as
        forward(Preprocess(frames)): padding to `size` (default conf.crop_size), /255, -mean, /stds, BGR->RGB
        (lib/augmentations.py:472-501, lib/dataloader.py:943-950) happen inside the stem kernel's loads."""
        if not frames.is_cuda or frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
            raise NotImplementedError("forward_u8: uint8 [B, h, w, 3] ROCm device tensor expected")
        H, W = (int(v) for v in (size if size is not None else self.conf.crop_size))
        B, h, w, _ = frames.shape
        if frames.device != self.device:
            raise RuntimeError("M3DSSD HIP engine on %s got frames on %s" % (self.device, frames.device))
        if h > H or w > W:
            raise RuntimeError("forward_u8: frame %dx%d does not fit the padded size %dx%d" % (h, w, H, W))
        plan = self.plan_for(B, H, W)
        frames = frames.contiguous()
        plan.named["input_u8"][:] = [frames.data_ptr(), h, w]
        try:
            return self._run_to(plan, fresh)
        finally:
            plan.named["input_u8"][0] = 0

    def forward_backbone(self, x):
        """Backbone + DCN up-sampling only (DLASeg.forward): returns the NHWC View of the 128-channel map."""
        B, _, H, W = x.shape
        plan = self.plan_for(B, H, W)
        x = x.contiguous()
        plan.named["input_ptr"][0] = x.data_ptr()
        self.run_plan(plan)
        return plan.named["feats0"]

    def run_plan(self, plan, start=0, end=None):
        """Issue plan.ops[start:end] on the ENGINE device's current stream (the whole forward by default); the engine's
        device is made current for the launches, whatever the caller's current device is."""
        with torch.cuda.device(self.device):
            self._run_plan(plan, start, end)

    def _run_plan(self, plan, start, end):
        st = _Stream.current(self.device)
        ops = plan.ops[start:end]
        if self.profile is None:
            brs = [br for br in getattr(plan, "branches", ()) if lo_hi_contains(start, end, len(plan.ops), br)]
            if brs:
                # side branches: ops [b0, b1) run on a second stream beside ops [b1, join) (no data dependence between the two
                # groups: the plan builder vouches for that); fork / join by stream waits, so the pattern is captured by a hipGraph
                lo, hi = start, (len(plan.ops) if end is None else end)
                main = torch.cuda.current_stream(self.device)
                if getattr(self, "_side", None) is None:
                    self._side = torch.cuda.Stream(self.device)
                side = self._side
                sst = ctypes.c_void_p(side.cuda_stream)
                forks = {br[0] for br in brs}
                joins = {br[2] for br in brs}
                on_side = set()
                for b0, b1, _ in brs:
                    on_side.update(range(b0, b1))
                for i in range(lo, hi):
                    if i in joins:
                        main.wait_stream(side)
                    if i in forks:
                        side.wait_stream(main)
                    plan.ops[i][3](sst if i in on_side else st)
                if hi in joins:
                    main.wait_stream(side)
                return
            for op in ops:
                op[3](st)
            return
        L = self.L
        evs = []
        for op in ops:
            if self.profile_kinds is not None and op[1] not in self.profile_kinds:
                op[3](st)
                continue
            e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
            _hip.check(L.m3d_event_create(ctypes.byref(e0)))
            _hip.check(L.m3d_event_create(ctypes.byref(e1)))
            _hip.check(L.m3d_event_record(e0, st))
            op[3](st)
            _hip.check(L.m3d_event_record(e1, st))
            evs.append((op, e0, e1))
        self._pending += evs

    def flush_profile(self):
        """Resolve the recorded HIP events (synchronises) into self.profile rows (name, kind, flops, ms)."""
        L = self.L
        for op, e0, e1 in self._pending:
            ms = ctypes.c_float()
            _hip.check(L.m3d_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
            self.profile.append((op[0], op[1], op[2], ms.value))
            L.m3d_event_destroy(e0)
            L.m3d_event_destroy(e1)
        self._pending = []
