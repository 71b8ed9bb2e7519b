"""forward() bodies of the sub-modules when they are called on their own (NCHW torch tensors in and
out, like the reference).  Each one converts to NHWC once, runs the same HIP launches the engine uses
for that stage, and converts back.  Parameters are re-packed per call (these entry points are for
drop-in use and tests; the whole-network path is m3dssd_amd/engine.py)."""
import ctypes

import numpy as np
import torch

from .. import _hip
from ..engine import BN_EPS, ConvDesc, Engine, View, _rup
from .ops import _require_cuda, _stream


def _to_nhwc(x, cpad_to=16):
    x = x.contiguous().float()
    n, c, h, w = x.shape
    cs = _rup(c, cpad_to)
    t = (torch.zeros if cs != c else torch.empty)(n * h * w * cs, device=x.device, dtype=torch.float32)
    _hip.check(_hip.lib().m3d_nchw_to_nhwc(x.data_ptr(), t.data_ptr(), n, c, h, w, cs, _stream()))
    return View(t, n, h, w, cs, cs), c


def _to_nchw(v, c):
    out = torch.empty(v.n, c, v.h, v.w, device=v.t.device, dtype=torch.float32)
    _hip.check(_hip.lib().m3d_nhwc_to_nchw(v.ptr, v.cs, out.data_ptr(), v.n, c, v.h, v.w, _stream()))
    return out


def _pack(weight, cin_pad, cout_pad_to):
    w = weight.detach().contiguous().float()
    co, ci, kh, kw = w.shape
    cop = _rup(co, cout_pad_to)
    wp = torch.empty(cop * kh * kw * cin_pad, device=w.device, dtype=torch.float32)
    _hip.check(_hip.lib().m3d_pack_conv_weight(w.data_ptr(), wp.data_ptr(), co, cop, ci, cin_pad, kh, kw, _stream()))
    return wp, co, cop, kh, kw


def _affine(co, bias, bn, dev):
    scale = torch.ones(co, device=dev)
    shift = torch.zeros(co, device=dev) if bias is None else bias.detach().float().clone()
    if bn is not None:
        s = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
        shift = (shift - bn.running_mean.detach()) * s + bn.bias.detach()
        scale = s
    return scale.contiguous().float(), shift.contiguous().float()


def conv_nhwc(v, weight, bias=None, bn=None, stride=1, pad=0, act=0, res=None, res_mode=0, sigmoid_from=-1, om=None,
              cout_pad_to=32, out_cs_to=4, wino=False, splitk=True, wino_variant=-1, wino_splitk=False, wino44=False, wino44_nb=0):
    """One m3d_conv2d_forward (or, with wino=True, m3d_wino_conv3x3_forward) launch on an NHWC view;
    returns (View, keepalive)."""
    wp, co, cop, kh, kw = _pack(weight, v.c, (128 if wino44_nb == 2 else 64) if wino44 else cout_pad_to)
    if wino44:
        from ..engine import pack_wino44
        assert weight.shape[1] == v.c, "wino44 path: no channel padding"
        wp = pack_wino44(weight, cop, v.t.device)
    elif wino:
        from ..engine import pack_wino
        assert weight.shape[1] == v.c, "wino path: no channel padding"
        wp = pack_wino(weight, cop, v.t.device)
    scale, shift = _affine(co, bias, bn, v.t.device)
    ho = (v.h + 2 * pad - kh) // stride + 1
    wo = (v.w + 2 * pad - kw) // stride + 1
    cs = _rup(co, out_cs_to)
    out_t = (torch.zeros if cs != co else torch.empty)(v.n * ho * wo * cs, device=v.t.device, dtype=torch.float32)
    out = View(out_t, v.n, ho, wo, cs, cs)
    d = ConvDesc()
    d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = v.ptr, v.cs, v.n, v.h, v.w, v.c
    d.wgt, d.Cout, d.Cout_pad = wp.data_ptr(), co, cop
    d.kh, d.kw, d.stride, d.pad, d.dil, d.Ho, d.Wo = kh, kw, stride, pad, 1, ho, wo
    d.out, d.out_cs = out.ptr, out.cs
    d.scale, d.shift = scale.data_ptr(), shift.data_ptr()
    if res is not None:
        d.res, d.res_cs, d.res_mode = res.ptr, res.cs, res_mode
    d.act, d.sigmoid_from = act, sigmoid_from
    if om is not None:
        d.dcn_offmask, d.dcn_om_cs = om.ptr, om.cs
    ws = None
    if wino44:
        if wino_splitk:                        # K slices as gridDim.z where the plan asks for them (small maps)
            splits, ws_bytes = ctypes.c_int(), ctypes.c_longlong()
            _hip.check(_hip.lib().m3d_wino44_splitk_plan(ctypes.byref(d), ctypes.byref(splits), ctypes.byref(ws_bytes)))
            if splits.value > 1:
                ws = torch.empty(ws_bytes.value // 4, device=v.t.device, dtype=torch.float32)
                d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws_bytes.value
        _hip.check(_hip.lib().m3d_wino44_conv3x3_forward_ex(ctypes.byref(d), wino44_nb, _stream()))
        return out, (wp, scale, shift, ws)
    if not wino:                               # small-M layers: give the igemm its split-K scratch
        splits, ws_bytes = ctypes.c_int(), ctypes.c_longlong()
        _hip.check(_hip.lib().m3d_conv2d_splitk_plan(ctypes.byref(d), ctypes.byref(splits), ctypes.byref(ws_bytes)))
        if splits.value > 1 and splitk:
            ws = torch.empty(ws_bytes.value // 4, device=v.t.device, dtype=torch.float32)
            d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws_bytes.value
    if wino:
        if wino_splitk:                        # workspace for the wave kernel's split-K form (4 splits at most)
            ws = torch.empty(4 * v.n * ho * wo * cop, device=v.t.device, dtype=torch.float32)
            d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel() * 4
        _hip.check(_hip.lib().m3d_wino_conv3x3_forward_ex(ctypes.byref(d), wino_variant, _stream()))
    else:
        _hip.check(_hip.lib().m3d_conv2d_forward(ctypes.byref(d), _stream()))
    return out, (wp, scale, shift, ws)


def dcn_layer_nhwc(dcn, v, bn=None, act=0):
    kk = dcn.kernel_size[0] * dcn.kernel_size[1]
    if dcn.deformable_groups != 1 or dcn.dilation != 1:
        raise NotImplementedError("DCN: deformable_groups=1, dilation=1 only")
    om, k1 = conv_nhwc(v, dcn.conv_offset_mask.weight, dcn.conv_offset_mask.bias, None, dcn.stride, dcn.padding,
                       sigmoid_from=2 * kk)
    out, k2 = conv_nhwc(v, dcn.weight, dcn.bias, bn, dcn.stride, dcn.padding, act=act, om=om, cout_pad_to=64)
    return out, (k1, k2, om)


def dcn_layer_forward(dcn, x):
    """DCN.forward, model/DCNv2/dcn_v2.py:64-70."""
    _require_cuda(x)
    with torch.no_grad(), torch.cuda.device(x.device):
        v, _ = _to_nhwc(x, 32)
        if dcn.deformable_groups == 1 and dcn.dilation == 1:       # the M3DSSD layers: everything stays NHWC
            out, keep = dcn_layer_nhwc(dcn, v)
            return _to_nchw(out, dcn.out_channels)
        # general module contract (model/DCNv2/test.py:169-179: deformable_groups = 2): offset / mask conv on the fused conv
        # kernel, then the drop-in op with the group count.  chunk(out, 3) + cat(o1, o2) of dcn_v2.py:66-67 = the first
        # 2*G*kk channels are the offsets, the last G*kk the (sigmoid) mask.
        kk, g = dcn.kernel_size[0] * dcn.kernel_size[1], dcn.deformable_groups
        om, keep = conv_nhwc(v, dcn.conv_offset_mask.weight, dcn.conv_offset_mask.bias, None, dcn.stride, dcn.padding,
                             sigmoid_from=2 * kk * g)
        om = _to_nchw(om, 3 * kk * g)
        from . import ops
        return ops.dcn_v2_forward(x.float().contiguous(), om[:, :2 * kk * g].contiguous(), om[:, 2 * kk * g:].contiguous(),
                                  dcn.weight.detach().float().contiguous(), dcn.bias.detach().float(), dcn.stride,
                                  dcn.padding, dcn.dilation, g)


def deform_conv_forward(mod, x):
    """DeformConv.forward, model/pose_dla_dcn.py:482-485 (eval-mode BatchNorm)."""
    _require_cuda(x)
    if mod.training:
        raise NotImplementedError("DeformConv: eval mode only (BatchNorm statistics are folded)")
    with torch.no_grad(), torch.cuda.device(x.device):
        v, _ = _to_nhwc(x, 32)
        out, keep = dcn_layer_nhwc(mod.conv, v, bn=mod.actf[0], act=1)
        return _to_nchw(out, mod.conv.out_channels)


def dlaseg_forward(mod, x):
    """DLASeg.forward, model/pose_dla_dcn.py:687-696 -> [B, 128, H/8, W/8]."""
    _require_cuda(x)
    if mod.training:
        raise NotImplementedError("DLASeg: eval mode only")
    ver = tuple((p.data_ptr(), p._version) for p in list(mod.parameters()) + list(mod.buffers()))
    if mod._engine is None or mod._engine[0] != ver:
        sd = {"base." + k: v for k, v in mod.state_dict().items()}
        mod._engine = (ver, Engine(sd, mod._conf, device=x.device, backbone_only=True))
    eng = mod._engine[1]
    with torch.no_grad():
        feats0 = eng.forward_backbone(x.float())
        return _to_nchw(feats0, feats0.c)


def _select_from_prob(prob):
    """prob [B, A, H, W] -> (idx int32 [B*HW], val float [B*HW]) via the same kernel as the engine
    (lowest index among equal maxima)."""
    B, A, H, W = prob.shape
    L = _hip.lib()
    idx = torch.empty(B * H * W, device=prob.device, dtype=torch.int32)
    val = torch.empty(B * H * W, device=prob.device, dtype=torch.float32)
    p = prob.contiguous().float()
    _hip.check(L.m3d_fg_top1(p.data_ptr(), B, A, H * W, idx.data_ptr(), val.data_ptr(), _stream()))
    return idx, val, p


def shape_align_forward(mod, x, prob):
    """shape_align.forward, model/module/feturealign_mgpu.py:153-208."""
    _require_cuda(x, prob)
    L = _hip.lib()
    with torch.no_grad(), torch.cuda.device(x.device):
        B, C, H, W = x.shape
        kk = mod.kernel_size[0] * mod.kernel_size[1]
        idx, val, _ = _select_from_prob(prob)
        tab = mod.offset_table.to(x.device).contiguous()
        om_cs = _rup(3 * kk, 4)
        om_t = torch.empty(B * H * W * om_cs, device=x.device, dtype=torch.float32)
        om = View(om_t, B, H, W, 3 * kk, om_cs)
        _hip.check(L.m3d_align_offsets(0, idx.data_ptr(), val.data_ptr(), float(mod.thresh), tab.data_ptr(), None, None,
                                       None, 0.0, 1.0, 0.0, 1.0, om.ptr, om.cs, B, mod.num_anchors, H * W, kk, 0,
                                       _stream()))
        v, _ = _to_nhwc(x, 32)
        out, keep = conv_nhwc(v, mod.align.weight, mod.align.bias, None, 1, mod.align.padding, act=0, res=v, om=om,
                              cout_pad_to=64)
        return _to_nchw(out, C)


def center_align_forward(mod, x, bbox_x, bbox_y, prob):
    """center_align.forward, model/module/feturealign_mgpu.py:48-99."""
    _require_cuda(x, bbox_x, bbox_y, prob)
    L = _hip.lib()
    with torch.no_grad(), torch.cuda.device(x.device):
        B, C, H, W = x.shape
        A = mod.num_anchors
        idx, val, _ = _select_from_prob(prob)
        bx, by = bbox_x.contiguous().float(), bbox_y.contiguous().float()
        awh = torch.stack([mod.anchors_w.view(-1), mod.anchors_h.view(-1)], 1).to(x.device).contiguous()
        om_t = torch.empty(B * H * W * 4, device=x.device, dtype=torch.float32)
        om = View(om_t, B, H, W, 3, 4)
        _hip.check(L.m3d_align_offsets(1, idx.data_ptr(), val.data_ptr(), float(mod.thresh), None, bx.data_ptr(),
                                       by.data_ptr(), awh.data_ptr(), float(mod.xy_mean[0]), float(mod.xy_std[0]),
                                       float(mod.xy_mean[1]), float(mod.xy_std[1]), om.ptr, om.cs, B, A, H * W, 1,
                                       A * H * W, _stream()))
        v, _ = _to_nhwc(x, 32)
        out, keep = conv_nhwc(v, mod.align.weight, mod.align.bias, None, 1, mod.align.padding, act=0, res=v, om=om,
                              cout_pad_to=64)
        return _to_nchw(out, C)


def anab_forward(mod, x):
    """ANAB.forward, model/module/attention.py:183-216 (x + softmax(Q K^) V^)."""
    _require_cuda(x)
    with torch.no_grad(), torch.cuda.device(x.device):
        v, _ = _to_nhwc(x)
        out = Engine.anab_standalone(mod, v)
        return _to_nchw(out, x.shape[1])
