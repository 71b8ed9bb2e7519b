"""Synthetic weights, configuration and frames.

The reference ships no trained weights, no KITTI data and no pickled ``conf``
(its .gitignore:6 drops data/), so benchmarks, smoke runs, golden fixtures and
parity tests all use a deterministic synthetic recipe (numpy PCG64, one stream per
tensor keyed by crc32 of its state_dict name, so values do not depend on order):

* ``param_spec`` lists every state_dict entry of ``RPN(dla34)`` -- 542 tensors, the
  contract in SURVEY.md 8b; tools/gen_golden.py asserts it equals the reference
  model's own ``state_dict()`` keys and shapes.
* conv_offset_mask is re-randomised (the reference zero-inits it,
  model/DCNv2/dcn_v2.py:60-62, which would make every DCN degenerate), BN running
  stats are randomised, and the background logit bias is raised so that both
  branches of the ``fg > 0.5`` hard mask (feturealign_mgpu.py:62,164) occur; the ANAB query / key projections are scaled
  so that the 337-key softmax is not saturated (see ANAB_QK_GAIN below).
"""
import math
import zlib
from collections import OrderedDict

import numpy as np
import torch

from .config import Config
from . import rpn_util

HEADS = ["cls", "bbox_x", "bbox_y", "bbox_w", "bbox_h", "bbox_x3d", "bbox_y3d"]
HEADS_TAIL = ["bbox_z3d"]
HEADS_TAIL2 = ["bbox_w3d", "bbox_h3d", "bbox_l3d", "bbox_rY3d"]
BG_BIAS = 4.2
BOX_OUT_GAIN = 0.15
ANAB_QK_GAIN = 0.1


def _bn(spec, p, c):
    spec[p + ".weight"] = (c,)
    spec[p + ".bias"] = (c,)
    spec[p + ".running_mean"] = (c,)
    spec[p + ".running_var"] = (c,)
    spec[p + ".num_batches_tracked"] = ()


def _conv(spec, p, co, ci, k, bias):
    spec[p + ".weight"] = (co, ci, k, k)
    if bias:
        spec[p + ".bias"] = (co,)


def _block(spec, p, ci, co):
    _conv(spec, p + ".conv1", co, ci, 3, True)
    _bn(spec, p + ".bn1", co)
    _conv(spec, p + ".conv2", co, co, 3, True)
    _bn(spec, p + ".bn2", co)


def _tree(spec, p, levels, ci, co, level_root, root_dim=0):
    if root_dim == 0:
        root_dim = 2 * co
    if level_root:
        root_dim += ci
    if levels == 1:
        _block(spec, p + ".tree1", ci, co)
        _block(spec, p + ".tree2", co, co)
        _conv(spec, p + ".root.conv", co, root_dim, 1, False)
        _bn(spec, p + ".root.bn", co)
    else:
        _tree(spec, p + ".tree1", levels - 1, ci, co, False, 0)
        _tree(spec, p + ".tree2", levels - 1, co, co, False, root_dim + co)
    if ci != co:
        _conv(spec, p + ".project.0", co, ci, 1, False)
        _bn(spec, p + ".project.1", co)


def _deform(spec, p, ci, co):
    _bn(spec, p + ".actf.0", co)
    spec[p + ".conv.weight"] = (co, ci, 3, 3)
    spec[p + ".conv.bias"] = (co,)
    spec[p + ".conv.conv_offset_mask.weight"] = (27, ci, 3, 3)
    spec[p + ".conv.conv_offset_mask.bias"] = (27,)


def _ida(spec, p, o, chans):
    for i in range(1, len(chans)):
        _deform(spec, "%s.proj_%d" % (p, i), chans[i], o)
        spec["%s.up_%d.weight" % (p, i)] = (o, 1, 4, 4)
        _deform(spec, "%s.node_%d" % (p, i), o, o)


def _head(spec, p, ci, co, k0):
    _conv(spec, p + ".0", 256, ci, k0, True)
    _bn(spec, p + ".1", 256)
    _conv(spec, p + ".3", 256, 256, 1, True)
    _bn(spec, p + ".4", 256)
    _conv(spec, p + ".6", co, 256, 1, True)


def param_spec(num_anchors=36, num_classes=4):
    """OrderedDict name -> shape, in the reference's registration order."""
    s = OrderedDict()
    ch = [16, 32, 64, 128, 256, 512]
    b = "base.base"
    _conv(s, b + ".base_layer.0", ch[0], 3, 7, False)
    _bn(s, b + ".base_layer.1", ch[0])
    _conv(s, b + ".level0.0", ch[0], ch[0], 3, False)
    _bn(s, b + ".level0.1", ch[0])
    _conv(s, b + ".level1.0", ch[1], ch[0], 3, False)
    _bn(s, b + ".level1.1", ch[1])
    _tree(s, b + ".level2", 1, ch[1], ch[2], False)
    _tree(s, b + ".level3", 2, ch[2], ch[3], True)
    _tree(s, b + ".level4", 2, ch[3], ch[4], True)
    _tree(s, b + ".level5", 1, ch[4], ch[5], True)
    _ida(s, "base.dla_up.ida_0", 256, [256, 512])
    _ida(s, "base.dla_up.ida_1", 128, [128, 256, 256])
    _ida(s, "base.ida_up", 128, [128, 256])
    _head(s, "cls", 128, num_anchors * num_classes, 3)
    for h in HEADS[1:]:
        _head(s, h, 128, num_anchors, 1)
    for p in ("center_align2d", "center_align3d"):
        s[p + ".align.weight"] = (128, 128, 1, 1)
        s[p + ".align.bias"] = (128,)
    s["shape_align.align.weight"] = (128, 128, 3, 3)
    s["shape_align.align.bias"] = (128,)
    s["shape_align.proj.weight"] = (128, 256, 1, 1)
    _head(s, "bbox_z3d", 128, num_anchors, 1)
    s["bbox_z3d_gl.0.value_conv.weight"] = (128, 128, 1, 1)
    s["bbox_z3d_gl.0.spatial_conv.weight"] = (4, 128, 1, 1)
    s["bbox_z3d_gl.0.key_conv.weight"] = (168, 128, 1, 1)
    s["bbox_z3d_gl.0.query_conv.weight"] = (168, 128, 1, 1)
    _bn(s, "bbox_z3d_gl.1", 128)
    for h in HEADS_TAIL2:
        _head(s, h, 128, num_anchors, 1)
    return s


def _rng(seed, name):
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def _bilinear_up(shape):
    c, _, k, _ = shape
    f = math.ceil(k / 2)
    cc = (2 * f - 1 - f % 2) / (2.0 * f)
    w = np.zeros(shape, dtype=np.float32)
    for i in range(k):
        for j in range(k):
            w[:, 0, i, j] = (1 - math.fabs(i / f - cc)) * (1 - math.fabs(j / f - cc))
    return w


def synth_state_dict(seed=0, num_anchors=36, num_classes=4):
    spec = param_spec(num_anchors, num_classes)
    sd = OrderedDict()
    for name, shape in spec.items():
        g = _rng(seed, name)
        prefix, leaf = name.rsplit(".", 1)
        is_bn = (prefix + ".running_var") in spec
        is_dcn_main = name.endswith(".conv.weight") and (prefix + ".conv_offset_mask.weight") in spec
        if leaf == "num_batches_tracked":
            v = np.zeros((), dtype=np.int64)
        elif leaf == "running_mean":
            v = g.normal(0.0, 0.1, shape)
        elif leaf == "running_var":
            v = g.uniform(0.8, 1.2, shape)
        elif is_bn and leaf == "weight":
            v = g.uniform(0.8, 1.2, shape)
        elif is_bn and leaf == "bias":
            v = g.normal(0.0, 0.1, shape)
        elif leaf == "bias":
            v = g.normal(0.0, 0.05, shape)
            if "conv_offset_mask" in name:
                v = g.normal(0.0, 0.5, shape)
            if name == "cls.6.bias":                          # class-major: channel = cls*A + a
                v = g.normal(0.0, 0.5, shape)
                v[:num_anchors] += BG_BIAS
        elif ".up_" in name:
            v = _bilinear_up(shape) * g.uniform(0.9, 1.1, (shape[0], 1, 1, 1))
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            if "conv_offset_mask" in name:
                v = g.normal(0.0, 0.7 / math.sqrt(fan_in), shape)
            elif is_dcn_main or name.endswith(".align.weight"):
                stdv = 3.74 / math.sqrt(fan_in)               # uniform like DCNv2.reset_parameters, gain tuned so features keep O(1) spatial variance
                v = g.uniform(-stdv, stdv, shape)
            elif name.startswith("bbox_z3d_gl.0"):
                v = g.normal(0.0, 1.0 / math.sqrt(fan_in), shape)
                if name.endswith(("query_conv.weight", "key_conv.weight")):
                    # unit-gain random query / key projections of O(10) features give attention logits of +-2800 over the 337
                    # keys: a saturated (one-hot) softmax whose output flips on fp32 roundoff -- the float64 run of the
                    # oracle graph then differs from its own float32 run by 6e-4 in z3d (tools/truth_probe.py), and the
                    # P.V product is hardly exercised.  0.1 on each side puts the logits at std ~2, max ~25 (mean top weight
                    # 0.3 at 1280x384): a working attention block, conditioned like a trained one.
                    v = v * ANAB_QK_GAIN
            else:
                v = g.normal(0.0, 0.8 * math.sqrt(2.0 / fan_in), shape)
                if name.startswith("bbox_") and name.endswith(".6.weight"):
                    v = v * BOX_OUT_GAIN                      # regression deltas ~N(0, 0.3): realistic align offsets
        dt = torch.int64 if leaf == "num_batches_tracked" else torch.float32
        sd[name] = torch.from_numpy(np.asarray(v)).to(dt).reshape(shape)
    return sd


def synth_conf(crop_size=(384, 1280), seed=0, batch_size=1, device="cuda:0"):
    """Stand-in for the pickled training conf: 2-D anchors from the reference recipe
    (lib/rpn_util.py:39-52,167-183), seeded 3-D anchor columns and bbox_means/stds."""
    rng = np.random.Generator(np.random.PCG64([seed, 0xC0F]))
    conf = Config()
    conf.crop_size = list(crop_size)
    conf.batch_size = batch_size
    conf.device = device
    a2d = rpn_util.generate_anchors_2d(conf.anchor_scales, conf.anchor_ratios, conf.feat_stride)
    n = a2d.shape[0]
    a3d = np.zeros((n, 5), dtype=np.float32)
    a3d[:, 0] = rng.uniform(5.0, 60.0, n)
    a3d[:, 1] = rng.normal(1.6, 0.1, n)
    a3d[:, 2] = rng.normal(1.5, 0.1, n)
    a3d[:, 3] = rng.normal(3.9, 0.3, n)
    a3d[:, 4] = rng.uniform(-1.0, 1.0, n)
    conf.anchors = np.concatenate([a2d, a3d], axis=1).astype(np.float32)
    conf.bbox_means = rng.normal(0.0, 0.05, (1, 11)).astype(np.float32)
    conf.bbox_stds = rng.uniform(0.1, 0.6, (1, 11)).astype(np.float32)
    return conf


def synth_frames(batch, crop_size=(384, 1280), seed=1234, pad_right_third=False):
    """randn frames (~ normalised-image statistics); optionally the right
    third zeroed, mimicking the test-time Padding (lib/augmentations.py:135-160)."""
    g = np.random.Generator(np.random.PCG64([seed, batch, crop_size[0], crop_size[1]]))
    x = g.standard_normal((batch, 3, crop_size[0], crop_size[1]), dtype=np.float32)
    if pad_right_third:
        x[:, :, :, (2 * crop_size[1]) // 3:] = 0.0
    return torch.from_numpy(x)


def synth_boxes(n, seed=0):
    """SURVEY.md 8d NMS micro-bench: centres U([0,1280]x[0,384]), w,h U(5,200)xU(5,150),
    distinct scores = permutation / n."""
    g = np.random.Generator(np.random.PCG64([seed, n, 0xB0]))
    c = g.uniform([0, 0], [1280, 384], (n, 2))
    wh = g.uniform([5, 5], [200, 150], (n, 2))
    s = (g.permutation(n).astype(np.float64) + 1.0) / n
    return np.concatenate([c - wh / 2, c + wh / 2, s[:, None]], axis=1).astype(np.float32)
