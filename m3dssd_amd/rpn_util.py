"""Host helpers the RPN module and the detector need from the reference's lib/rpn_util.py.

Only the functions on the inference path are provided:
  anchor_center / generate_anchors_2d  <- lib/rpn_util.py:167-183, :39-52
  calc_output_size                     <- lib/rpn_util.py:1401-1413
  locate_anchors                       <- lib/rpn_util.py:1329-1398
  flatten_tensor                       <- lib/rpn_util.py:892-901
Everything is computed in float64 numpy and converted with ``.float()`` last, exactly
like the reference, so ``rois`` is bit-identical (tests/test_oracle_golden.py).
"""
import numpy as np
import torch


def anchor_center(w, h, stride):
    half = (stride - 1) / 2
    return np.array([-w / 2 + half, -h / 2 + half, w / 2 + half, h / 2 + half], dtype=np.float32)


def generate_anchors_2d(scales, ratios, feat_stride):
    rows = [anchor_center(s * r, s, feat_stride) for s in scales for r in ratios]
    return np.stack(rows).astype(np.float32)


def calc_output_size(res, stride):
    return np.ceil(np.array(res) / stride).astype(int)


def locate_anchors(anchors, feat_size, stride, convert_tensor=False):
    """[(A*H*W), 5] = (x1, y1, x2, y2, anchor index); row = (a*H + h)*W + w."""
    if torch.is_tensor(anchors):
        anchors = anchors.detach().cpu().numpy()
    H, W = int(feat_size[0]), int(feat_size[1])
    xs = np.arange(W, dtype=np.float64) * float(stride)
    ys = np.arange(H, dtype=np.float64) * float(stride)
    box = anchors[:, 0:4]
    A = box.shape[0]
    rois = np.empty((A, H, W, 5), dtype=np.float64)
    rois[..., 0] = xs[None, None, :] + box[:, 0][:, None, None]
    rois[..., 1] = ys[None, :, None] + box[:, 1][:, None, None]
    rois[..., 2] = xs[None, None, :] + box[:, 2][:, None, None]
    rois[..., 3] = ys[None, :, None] + box[:, 3][:, None, None]
    rois[..., 4] = np.arange(A, dtype=np.float64)[:, None, None]
    rois = rois.reshape(A * H * W, 5)
    return torch.from_numpy(rois) if convert_tensor else rois


def flatten_tensor(t):
    """[B, C, H, W] -> [B, H*W, C]."""
    b, c = t.shape[0], t.shape[1]
    return t.permute(0, 2, 3, 1).contiguous().view(b, -1, c)
