"""``test_kitti_3d`` with the reference's signature (lib/rpn_util.py:1754-1893): run the detector over a test set, refine
every kept box (alpha -> ry, hill climb, back-projection), write one KITTI result file per image, evaluate AP against the
label folder.

What changes on the MI355X path: the reference handles ONE image per iteration (forward, host decode / sort / NMS, a Python
loop of <= 40 boxes x ~14 projections); here frames are grouped into batches of ``rpn_conf.batch_size`` and each batch is
forward -> decode -> top-k -> NMS -> select (``detect_batch``) -> ``m3d_refine_3d`` on the device; the host only formats text.
The result files and the printed AP tables are the reference's, byte for byte, given the same detections.
"""
import logging
import os

import numpy as np
import torch

from ..eval import get_label_annos, get_official_eval_result
from . import refine as R
from .detect import detect_batch


def _unpack(batch, rpn_conf):
    """One item of the reference's test loader -> (image tensor [1,3,H,W] or [3,H,W], meta object with .id/.p2/...)."""
    if getattr(rpn_conf, "pre_compute_target", False) or isinstance(batch, dict):
        meta = batch["target"]["meta"]
        return batch["input"], meta
    return batch


def _get(obj, key, default=None):
    if isinstance(obj, dict):
        return obj.get(key, default)
    return getattr(obj, key, default)


def _flush(ims, metas, net, rpn_conf, results_path, dev):
    x = torch.cat([im if im.dim() == 4 else im[None] for im in ims]).to(dev, torch.float32)
    dets, counts = detect_batch(net, x, rpn_conf)
    dets = dets.clone()
    B = dets.shape[0]
    scale = torch.tensor([float(_get(m, "scale_factor", 1.0) or 1.0) for m in metas], device=dev, dtype=torch.float32)
    if bool((scale != 1.0).any()):                      # lib/rpn_util.py:1528-1531: back to the original image scale
        dets[:, :, 0:4] /= scale[:, None, None]
        dets[:, :, 6:8] /= scale[:, None, None]
    if getattr(rpn_conf, "clip_boxes", False):          # :1533-1538
        for b, m in enumerate(metas):
            w, h = _get(m, "imW"), _get(m, "imH")
            if w is not None and h is not None:
                dets[b, :, 0].clamp_(0, w - 1)
                dets[b, :, 2].clamp_(0, w - 1)
                dets[b, :, 1].clamp_(0, h - 1)
                dets[b, :, 3].clamp_(0, h - 1)
    p2 = np.stack([np.asarray(_get(m, "p2"), dtype=np.float64).reshape(4, 4) for m in metas])
    ref = R.refine_detections(dets, counts, p2, hill_climbing=bool(getattr(rpn_conf, "hill_climbing", True))).cpu().numpy()
    for b, m in enumerate(metas):
        with open(os.path.join(results_path, str(_get(m, "id")) + ".txt"), "w") as f:
            f.write(R.kitti_text(ref[b], rpn_conf.lbls))
    return B


def test_kitti_3d(dataset_test, net, rpn_conf, results_path, test_path, use_log=True, writer=None, phase="validation"):
    """Same arguments as the reference.  Returns (result text, stats dict) of get_official_eval_result, or (None, None) when
    the label folder of `phase` does not exist (the reference would raise while reading it)."""
    os.makedirs(results_path, exist_ok=True)
    dev = next(net.parameters()).device
    net.eval()
    bs = max(1, int(getattr(rpn_conf, "batch_size", 1)))
    ims, metas, n_done = [], [], 0
    with torch.no_grad():
        for batch in dataset_test:
            im, meta = _unpack(batch, rpn_conf)
            im = torch.as_tensor(im)
            if ims and tuple(im.shape[-2:]) != tuple(ims[0].shape[-2:]):      # a different padded size: finish the open batch
                n_done += _flush(ims, metas, net, rpn_conf, results_path, dev)
                ims, metas = [], []
            ims.append(im)
            metas.append(meta)
            if len(ims) == bs:
                n_done += _flush(ims, metas, net, rpn_conf, results_path, dev)
                ims, metas = [], []
        if ims:
            n_done += _flush(ims, metas, net, rpn_conf, results_path, dev)
    key = "datasets_{}".format(phase)
    sub = {"validation": "validation", "train": "training", "test": "testing"}[phase]
    gt_path = None
    if key in rpn_conf and rpn_conf[key]:
        gt_path = os.path.join(test_path, _get(rpn_conf[key][0], "name"), sub, "label_2")
    if gt_path is None or not os.path.isdir(gt_path):
        return None, None
    dt_annos = get_label_annos(results_path)
    gt_annos = get_label_annos(gt_path)
    res, res_stats = get_official_eval_result(gt_annos, dt_annos, [0, 1, 2])
    if writer is not None:                                 # lib/rpn_util.py:1880-1896
        for lbl in rpn_conf.lbls:
            for item in ("aos", "3d", "bev", "image"):
                k = "{}_{}".format(lbl, item)
                if k + "_easy" in res_stats:
                    writer.add_scalars("Test/" + k, {"easy": res_stats[k + "_easy"], "mod": res_stats[k + "_moderate"],
                                                     "hard": res_stats[k + "_hard"]}, 0)
    if use_log:
        logging.info(res)
    else:
        print(res)
    return res, res_stats
