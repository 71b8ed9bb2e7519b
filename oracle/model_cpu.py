"""ORACLE (test infrastructure): the whole per-image forward on the CPU.

A functional (state_dict-driven) restatement of the reference graph using torch's
CPU primitives -- the same L0 primitives (conv2d / batch_norm / leaky_relu /
max_pool2d / conv_transpose2d / softmax / topk / gather / adaptive_avg_pool2d /
bmm) the reference itself calls -- plus oracle/dcn.py for the DCNv2 op:

  RPN.forward ............ model/M3d_inference_align.py:215-313
  DLASeg.forward ......... model/pose_dla_dcn.py:687-696
  DLA.forward / dla34 .... model/pose_dla_dcn.py:391-397, 419-425
  Tree / Root / BasicBlock model/pose_dla_dcn.py:314-327, 261-269, 107-121
  DLAUp / IDAUp .......... model/pose_dla_dcn.py:572-578, 546-552
  DeformConv / DCN ....... model/pose_dla_dcn.py:482-485, model/DCNv2/dcn_v2.py:64-70
  shape_align ............ model/module/feturealign_mgpu.py:153-208 (table :119-136)
  center_align ........... model/module/feturealign_mgpu.py:48-99
  ANAB / PAPAModule ...... model/module/attention.py:183-216, 136-147
  flatten_tensor ......... lib/rpn_util.py:892-901

``sd`` uses the reference's state_dict keys.  ``taps`` (optional
dict) receives named intermediates; ``inject`` may carry discrete decisions
(top-1 anchor index / hard mask) taken from another run so that stage-wise
comparisons are not derailed by 1-ulp flips (SURVEY.md 7 "hard parts").
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import anchors as A
from . import dcn as D

BN_EPS = 1e-5
SLOPE = 0.01  # nn.LeakyReLU default negative_slope


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def _conv(sd, p, x, stride=1, pad=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=pad)


def _lrelu(x):
    return F.leaky_relu(x, SLOPE)


def _basic_block(sd, p, x, residual, stride):
    out = _lrelu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, stride, 1)))
    out = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, 1, 1))
    return _lrelu(out + residual)


def _root(sd, p, xs):
    return _lrelu(_bn(sd, p + ".bn", _conv(sd, p + ".conv", torch.cat(xs, 1))))


def _tree(sd, p, x, levels, stride, level_root, children=None):
    children = [] if children is None else children
    bottom = F.max_pool2d(x, stride, stride) if stride > 1 else x
    has_proj = (p + ".project.0.weight") in sd
    if level_root:
        children.append(bottom)
    if levels == 1:
        residual = _bn(sd, p + ".project.1", _conv(sd, p + ".project.0", bottom)) if has_proj else bottom
        x1 = _basic_block(sd, p + ".tree1", x, residual, stride)
        x2 = _basic_block(sd, p + ".tree2", x1, x1, 1)
        return _root(sd, p + ".root", [x2, x1] + children)
    # levels == 2: the outer project is computed by the reference but its result is
    # overwritten inside tree1.forward (pose_dla_dcn.py:317) -> no effect on outputs.
    x1 = _tree(sd, p + ".tree1", x, levels - 1, stride, False)
    children.append(x1)
    return _tree(sd, p + ".tree2", x1, levels - 1, 1, False, children)


def dla34(sd, p, x, taps=None):
    x = _lrelu(_bn(sd, p + ".base_layer.1", _conv(sd, p + ".base_layer.0", x, 1, 3)))
    x = _lrelu(_bn(sd, p + ".level0.1", _conv(sd, p + ".level0.0", x, 1, 1)))
    y = [x]
    x = _lrelu(_bn(sd, p + ".level1.1", _conv(sd, p + ".level1.0", x, 2, 1)))
    y.append(x)
    for lvl, (levels, level_root) in zip((2, 3, 4, 5), ((1, False), (2, True), (2, True), (1, True))):
        x = _tree(sd, "%s.level%d" % (p, lvl), x, levels, 2, level_root)
        y.append(x)
    if taps is not None:
        for i, t in enumerate(y):
            taps["level%d" % i] = t
    return y


def dcn_layer(sd, p, x, taps=None, name=None):
    """DCN.forward, model/DCNv2/dcn_v2.py:64-70 (3x3, stride 1, pad 1)."""
    om = _conv(sd, p + ".conv_offset_mask", x, 1, 1)
    o1, o2, m = torch.chunk(om, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1)
    mask = torch.sigmoid(m)
    if taps is not None and name:
        taps[name + ".offset"], taps[name + ".mask"] = offset, mask
    return D.dcn_v2_forward(x, offset, mask, sd[p + ".weight"], sd[p + ".bias"], 1, 1, 1, 1)


def deform_conv(sd, p, x, taps=None, name=None):
    y = _lrelu(_bn(sd, p + ".actf.0", dcn_layer(sd, p + ".conv", x, taps, name)))
    if taps is not None and name:
        taps[name + ".in"], taps[name + ".out"] = x, y
    return y


def _ida_up(sd, p, layers, startp, endp, taps=None):
    for i in range(startp + 1, endp):
        j = i - startp
        w = sd["%s.up_%d.weight" % (p, j)]
        c, f2 = w.shape[0], w.shape[2]
        f = f2 // 2
        t = deform_conv(sd, "%s.proj_%d" % (p, j), layers[i], taps, "%s.proj_%d" % (p, j))
        t = F.conv_transpose2d(t, w, None, stride=f, padding=f // 2, groups=c)
        layers[i] = deform_conv(sd, "%s.node_%d" % (p, j), t + layers[i - 1], taps, "%s.node_%d" % (p, j))


def dla_seg(sd, p, x, taps=None):
    """DLASeg.forward with down_ratio 8 (first_level 3), last_level 5."""
    layers = dla34(sd, p + ".base", x, taps)
    first = 3
    out = [layers[-1]]
    for i in range(len(layers) - first - 1):
        _ida_up(sd, "%s.dla_up.ida_%d" % (p, i), layers, len(layers) - i - 2, len(layers), taps)
        out.insert(0, layers[-1])
    y = [out[0].clone(), out[1].clone()]
    _ida_up(sd, p + ".ida_up", y, 0, len(y), taps)
    return y[-1]


def head(sd, p, x, k0=1):
    x = _lrelu(_bn(sd, p + ".1", _conv(sd, p + ".0", x, 1, k0 // 2)))
    x = _lrelu(_bn(sd, p + ".4", _conv(sd, p + ".3", x)))
    return _conv(sd, p + ".6", x)


def anchor_select(fg_prob, thresh=0.5, inject=None):
    """topk(k=1) + max + hard mask, feturealign_mgpu.py:58-62 / :160-164.
    Tie rule (torch.topk's is unspecified): lowest anchor index wins."""
    mask, ind = torch.max(fg_prob, dim=1, keepdim=True)
    # torch.max returns the first max index on CPU for ties -> lowest index
    hard = (mask > thresh).float()
    if inject is not None:
        ind = inject.get("ind", ind)
        mask = torch.gather(fg_prob, 1, ind)
        hard = inject.get("hard", (mask > thresh).float())
    return mask, ind, hard


def shape_align_offsets(anchors, feat_stride, ks=3):
    """Per-anchor offset table, feturealign_mgpu.py:119-136 -> [A, 2*ks*ks]."""
    a = torch.as_tensor(anchors, dtype=torch.float32)
    aw = (a[:, 2] - a[:, 0])
    ah = (a[:, 3] - a[:, 1])
    h_step = ah / feat_stride / ks
    w_step = aw / feat_stride / ks
    tab = torch.zeros(a.shape[0], 2 * ks * ks, dtype=torch.float32)
    for i in range(ks):
        for j in range(ks):
            b = i * ks + j
            tab[:, 2 * b] = (h_step - 1) * (i - ks / 2 + 0.5)
            tab[:, 2 * b + 1] = (w_step - 1) * (j - ks / 2 + 0.5)
    return tab


def shape_align(sd, p, x, fg_prob, anchors, feat_stride, taps=None, inject=None):
    mask, ind, hard = anchor_select(fg_prob, 0.5, inject)
    tab = shape_align_offsets(anchors, feat_stride, 3)            # [A, 18]
    off = tab[ind[:, 0]].permute(0, 3, 1, 2).contiguous()         # [B,18,H,W]; softmax over k=1 == 1
    off = off * hard
    m9 = mask.repeat(1, 9, 1, 1)
    if taps is not None:
        taps["shape_align.ind"], taps["shape_align.hard"] = ind, hard
        taps["shape_align.offset"], taps["shape_align.mask"] = off, m9
    out = D.dcn_v2_forward(x, off, m9, sd[p + ".align.weight"], sd[p + ".align.bias"], 1, 1, 1, 1)
    return out + x


def center_align(sd, p, x, bbox_x, bbox_y, fg_prob, anchors, xy_mean, xy_std, feat_stride,
                 taps=None, name="center_align", inject=None):
    mask, ind, hard = anchor_select(fg_prob, 0.5, inject)
    a = torch.as_tensor(anchors, dtype=torch.float32)
    aw = ((a[:, 2] - a[:, 0]) / feat_stride).view(1, -1, 1, 1)
    ah = ((a[:, 3] - a[:, 1]) / feat_stride).view(1, -1, 1, 1)
    mx, my = torch.tensor(float(xy_mean[0])), torch.tensor(float(xy_mean[1]))
    sx, sy = torch.tensor(float(xy_std[0])), torch.tensor(float(xy_std[1]))
    off_x = (bbox_x * sx + mx) * aw
    off_y = (bbox_y * sy + my) * ah
    off_x = torch.gather(off_x, 1, ind) * hard
    off_y = torch.gather(off_y, 1, ind) * hard
    off = torch.cat([off_y, off_x], dim=1)                         # (dh, dw) for the single tap
    if taps is not None:
        taps[name + ".offset"], taps[name + ".mask"] = off, mask
    out = D.dcn_v2_forward(x, off, mask, sd[p + ".align.weight"], sd[p + ".align.bias"], 1, 0, 1, 1)
    return out + x


def anab(sd, p, x, psp=(1, 4, 8, 16), taps=None):
    B, C, H, W = x.shape
    q = _conv(sd, p + ".query_conv", x)
    kc = q.shape[1]
    q = q.view(B, kc, H * W).permute(0, 2, 1)
    s = torch.sigmoid(_conv(sd, p + ".spatial_conv", x))
    k = _conv(sd, p + ".key_conv", x)
    v = _conv(sd, p + ".value_conv", x)
    kp = torch.cat([F.adaptive_avg_pool2d(k * s[:, i:i + 1], (z, z)).view(B, kc, -1)
                    for i, z in enumerate(psp)], -1)
    vp = torch.cat([F.adaptive_avg_pool2d(v * s[:, i:i + 1], (z, z)).view(B, C, -1)
                    for i, z in enumerate(psp)], -1).permute(0, 2, 1)
    att = torch.softmax(torch.bmm(q, kp), dim=-1)
    nv = torch.bmm(att, vp).permute(0, 2, 1).reshape(B, C, H, W)
    if taps is not None:
        taps["anab.key_pooled"], taps["anab.value_pooled"] = kp, vp
    return (nv + x).contiguous()


def _flat(x):
    b, c = x.shape[0], x.shape[1]
    return x.permute(0, 2, 3, 1).contiguous().view(b, -1, c)


def rpn_forward(sd, conf, x, taps=None, inject=None):
    """-> cls, prob, bbox_2d, bbox_3d, feat_size, rois  (eval-mode outputs, :303-313)."""
    inject = inject or {}
    B = x.shape[0]
    anchors = np.asarray(conf.anchors, dtype=np.float32)
    na, nc = anchors.shape[0], len(conf.lbls) + 1
    means, stds = conf.bbox_means[0], conf.bbox_stds[0]
    feats0 = dla_seg(sd, "base", x, taps)
    fh, fw = feats0.shape[2], feats0.shape[3]
    cls = head(sd, "cls", feats0, 3).view(B, nc, fh * na, fw)
    prob = torch.softmax(cls, dim=1)
    fg = (1 - prob[:, 0]).view(B, na, fh, fw)
    feats = shape_align(sd, "shape_align", feats0, fg, anchors, conf.feat_stride, taps, inject.get("sel"))
    bx, by = head(sd, "bbox_x", feats), head(sd, "bbox_y", feats)
    f2d = center_align(sd, "center_align2d", feats, bx, by, fg, anchors, means[0:2], stds[0:2],
                       conf.feat_stride, taps, "center_align2d", inject.get("sel"))
    bw, bh = head(sd, "bbox_w", f2d), head(sd, "bbox_h", f2d)
    bx3, by3 = head(sd, "bbox_x3d", feats), head(sd, "bbox_y3d", feats)
    f3d = center_align(sd, "center_align3d", feats, bx3, by3, fg, anchors, means[4:6], stds[4:6],
                       conf.feat_stride, taps, "center_align3d", inject.get("sel"))
    bw3, bh3 = head(sd, "bbox_w3d", f3d), head(sd, "bbox_h3d", f3d)
    bl3, br3 = head(sd, "bbox_l3d", f3d), head(sd, "bbox_rY3d", f3d)
    gl = _lrelu(_bn(sd, "bbox_z3d_gl.1", anab(sd, "bbox_z3d_gl.0", f3d, taps=taps)))
    bz3 = head(sd, "bbox_z3d", gl)
    if taps is not None:
        taps.update({"feats0": feats0, "fg_prob": fg, "feats": feats, "feats_align2d": f2d,
                     "feats_align3d": f3d, "feats_gl": gl})
    fl = lambda t: _flat(t.view(B, 1, fh * na, fw))
    bbox_2d = torch.cat([fl(t) for t in (bx, by, bw, bh)], dim=2)
    bbox_3d = torch.cat([fl(t) for t in (bx3, by3, bz3, bw3, bh3, bl3, br3)], dim=2)
    feat_size = torch.tensor([fh, fw], dtype=torch.float)
    rois = torch.from_numpy(A.locate_anchors(anchors, [fh, fw], conf.feat_stride)).float()
    return _flat(cls), _flat(prob), bbox_2d, bbox_3d, feat_size, rois
