"""Post-NMS 3-D refinement and KITTI result formatting on the device.

The reference does this per box in Python after `im_detect_3d` (lib/rpn_util.py:1801-1852: convertAlpha2Rot, hill_climb,
convertRot2Alpha, the result line); here one kernel launch refines every row of a batch (`m3d_refine_3d`, float64) and the
host only formats text.  Row format in and out is the reference's (`aboxes` rows, lib/rpn_util.py:1550; result line,
lib/rpn_util.py:1848-1849).
"""
import ctypes
import math
import os

import numpy as np
import torch

from .. import _hip

LINE = ("{} -1 -1 {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} " + "{:.6f} {:.6f}\n")


def p2_arrays(p2, B):
    """p2 [B, 4, 4] or [4, 4] (host) -> (p2, inverse) as C-contiguous float64 [B, 4, 4]; the inverse is the reference's
    np.linalg.inv (lib/rpn_util.py:1790)."""
    p2 = np.asarray(p2, dtype=np.float64)
    if p2.ndim == 2:
        p2 = np.broadcast_to(p2, (B, 4, 4))
    if p2.shape != (B, 4, 4):
        raise RuntimeError("refine_detections: p2 must be [B, 4, 4]")
    return np.array(p2, dtype=np.float64, order="C"), np.ascontiguousarray(np.stack([np.linalg.inv(m) for m in p2]))


def refine_detections(dets, counts, p2, score_thresh=0.75, hill_climbing=True, step_r_init=0.3 * math.pi, r_lim=0.01,
                      scale=None, clip_wh=None):
    """dets [B, K, 14] float32 device rows (detect_batch / im_detect_3d format), counts [B] int32 device, p2 [B, 4, 4] (or
    [4, 4]) projection matrices (numpy / host) -> float64 device tensor [B, K, 16]:
    valid, cls, alpha, x1, y1, x2, y2, h3d, w3d, l3d, x3d, y3d, z3d, ry3d, score, 0.
    scale [B] (host floats): the rows are first divided back to the original image scale (lib/rpn_util.py:1506-1507; None when detect_batch(..., scale=) already did it before the NMS, the reference order);
    clip_wh [B, 2] = (imW, imH): the 2-D boxes are clipped to the image (:1533-1538; an entry <= 0 = not clipped)."""
    if not dets.is_cuda:
        raise NotImplementedError("refine_detections: ROCm device tensors expected")
    if dets.dim() != 3 or dets.shape[2] != 14 or dets.dtype != torch.float32:
        raise RuntimeError("refine_detections: dets must be float32 [B, K, 14]")
    B, K, _ = dets.shape
    p2, p2_inv = p2_arrays(p2, B)
    dev = dets.device
    d_p2 = torch.from_numpy(p2).to(dev)
    d_pi = torch.from_numpy(p2_inv).to(dev)
    d_sc = None if scale is None else torch.as_tensor(np.asarray(scale, dtype=np.float32).reshape(B)).to(dev)
    d_cl = None if clip_wh is None else torch.as_tensor(np.asarray(clip_wh, dtype=np.float32).reshape(B, 2)).to(dev)
    dets = dets.contiguous()
    counts = counts.to(device=dev, dtype=torch.int32).contiguous()
    out = torch.empty(B, K, 16, device=dev, dtype=torch.float64)
    with torch.cuda.device(dev):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _hip.check(_hip.lib().m3d_refine_3d_ex(dets.data_ptr(), counts.data_ptr(), B, K, d_p2.data_ptr(), d_pi.data_ptr(),
                                               None if d_sc is None else d_sc.data_ptr(),
                                               None if d_cl is None else d_cl.data_ptr(), float(score_thresh),
                                               1 if hill_climbing else 0, float(step_r_init), float(r_lim), out.data_ptr(), st))
    return out


def kitti_text(refined_rows, lbls):
    """refined rows of ONE image ([K, 16], host) -> the text the reference writes for it (lib/rpn_util.py:1848-1850)."""
    text = ""
    for r in np.asarray(refined_rows, dtype=np.float64):
        if r[0] != 0.0:
            text += LINE.format(lbls[int(r[1] - 1)], *[float(v) for v in r[2:15]])
    return text


def write_kitti_results(dets, counts, p2, ids, results_path, conf):
    """The result files of test_kitti_3d (lib/rpn_util.py:1796-1852) for a batch: one '<id>.txt' per image."""
    ref = refine_detections(dets, counts, p2, hill_climbing=bool(getattr(conf, "hill_climbing", True))).cpu().numpy()
    os.makedirs(results_path, exist_ok=True)
    for b, name in enumerate(ids):
        with open(os.path.join(results_path, str(name) + ".txt"), "w") as f:
            f.write(kitti_text(ref[b], conf.lbls))
    return ref
