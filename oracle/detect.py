"""ORACLE (test infrastructure): decode + top-k + NMS after the forward.

Restates lib/rpn_util.py:1442-1553 (im_detect_3d, the ``synced=False`` branch)
and bbox_transform_inv (:1137-1186), generalised from the reference's batch
index 0 (:1484-1503) to a per-image loop.  scale_factor is 1 (synthetic frames are
already at test scale).  Sort order: descending score, ascending row among equals
(the reference's torch.argsort(-score) is unstable; see oracle/nms.py).

Row format of the result (rpn_util.py:1550): x1,y1,x2,y2,score,cls,x3d,y3d,z3d,w3d,h3d,l3d,ry3d,anchor.
"""
import numpy as np
import torch

from . import nms as N


def decode(prob, bbox_2d, bbox_3d, rois, conf):
    """One image.  prob [R,4], bbox_2d [R,4], bbox_3d [R,7], rois [R,5] (float32 tensors).
    -> coords_2d [R,4], coords_3d [R,7], scores [R], cls_pred [R], tracker [R]."""
    anchors = torch.from_numpy(np.asarray(conf.anchors)).float()
    means = torch.from_numpy(np.asarray(conf.bbox_means)).float()[0]
    stds = torch.from_numpy(np.asarray(conf.bbox_stds)).float()[0]
    d3 = [bbox_3d[:, i] * stds[4 + i] + means[4 + i] for i in range(7)]
    tracker = rois[:, 4].long()
    src = anchors[tracker, 4:]
    widths = rois[:, 2] - rois[:, 0] + 1.0
    heights = rois[:, 3] - rois[:, 1] + 1.0
    ctr_x = rois[:, 0] + 0.5 * widths
    ctr_y = rois[:, 1] + 0.5 * heights
    x3d = d3[0] * widths + ctr_x
    y3d = d3[1] * heights + ctr_y
    z3d = src[:, 0] + d3[2]
    w3d = torch.exp(d3[3]) * src[:, 1]
    h3d = torch.exp(d3[4]) * src[:, 2]
    l3d = torch.exp(d3[5]) * src[:, 3]
    ry3d = src[:, 4] + d3[6]
    coords_3d = torch.stack((x3d, y3d, z3d, w3d, h3d, l3d, ry3d), dim=1)
    # bbox_transform_inv, :1137-1186
    dx = bbox_2d[:, 0] * stds[0] + means[0]
    dy = bbox_2d[:, 1] * stds[1] + means[1]
    dw = bbox_2d[:, 2] * stds[2] + means[2]
    dh = bbox_2d[:, 3] * stds[3] + means[3]
    pcx = dx * widths + ctr_x
    pcy = dy * heights + ctr_y
    pw = torch.exp(dw) * widths
    ph = torch.exp(dh) * heights
    coords_2d = torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=1)
    scores, am = torch.max(prob[:, 1:], dim=1)
    return coords_2d, coords_3d, scores, (am + 1).float(), tracker.float()


def detect_image(prob, bbox_2d, bbox_3d, rois, conf):
    """-> (aboxes [K,14] float32 ndarray, kept indices into the top-N-pre list, top-N-pre row ids)."""
    c2, c3, scores, cls_pred, tracker = decode(prob, bbox_2d, bbox_3d, rois, conf)
    order = torch.from_numpy(N.order_desc_stable(scores.numpy()))
    top = order[:min(conf.nms_topN_pre, order.shape[0])]
    ab = torch.cat((c2[top], scores[top, None]), dim=1)
    keep = N.gpu_nms(ab.numpy().astype(np.float32), conf.nms_thres)
    full = torch.cat((ab, cls_pred[top, None], c3[top], tracker[top, None]), dim=1)
    return full[keep].numpy(), np.asarray(keep, dtype=np.int64), top.numpy()
