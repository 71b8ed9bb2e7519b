"""``test_kitti_3d`` with the reference's signature (lib/rpn_util.py:1754-1893): run the detector over a test set, refine
every kept box (alpha -> ry, hill climb, back-projection), write one KITTI result file per image, evaluate AP against the
label folder.

What changes on the MI355X path: the reference handles ONE image per iteration (forward, host decode / sort / NMS, a Python
loop of <= 40 boxes x ~14 projections); here frames are grouped into batches of ``rpn_conf.batch_size`` and each batch is
forward -> decode -> top-k -> NMS -> select -> ``m3d_refine_3d_ex`` on the device, full batches inside ONE captured hipGraph
(forward of batch k beside the detection + refinement of batch k-1); the host only formats text.
The result files and the printed AP tables are the reference's, byte for byte, given the same detections.
"""
import logging
import os

import numpy as np
import torch

from ..eval import get_label_annos, get_official_eval_result
from . import refine as R
from .detect import detect_batch, unwrap


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


def _meta_arrays(metas, rpn_conf):
    """Per-image calibration / scale / clip size of a batch, as m3d_refine_3d_ex takes them."""
    p2 = np.stack([np.asarray(_get(m, "p2"), dtype=np.float64).reshape(4, 4) for m in metas])
    scale = np.asarray([float(_get(m, "scale_factor", 1.0) or 1.0) for m in metas], dtype=np.float32)   # rpn_util.py:1435,1506-1507
    clip = np.zeros((len(metas), 2), dtype=np.float32)
    if getattr(rpn_conf, "clip_boxes", False):                                                           # :1533-1538
        for b, m in enumerate(metas):
            w, h = _get(m, "imW"), _get(m, "imH")
            if w is not None and h is not None:
                clip[b] = (w, h)
    return {"p2": p2, "scale": scale, "clip_wh": clip}


def _write(refined, metas, rpn_conf, results_path):
    ref = refined.cpu().numpy()
    for b, m in enumerate(metas):
        with open(os.path.join(results_path, str(_get(m, "id")) + ".txt"), "w") as f:
            f.write(R.kitti_text(ref[b], rpn_conf.lbls))


def _flush(ims, metas, net, rpn_conf, results_path, dev):
    """A batch outside the pipelined graph (the ragged last one, or a one-off padded size): eager launches."""
    x = torch.cat([im if im.dim() == 4 else im[None] for im in ims]).to(dev, torch.float32)
    meta = _meta_arrays(metas, rpn_conf)
    dets, counts = detect_batch(net, x, rpn_conf, scale=meta["scale"])          # scaled before the NMS, like the reference
    ref = R.refine_detections(dets, counts, meta["p2"], hill_climbing=bool(getattr(rpn_conf, "hill_climbing", True)),
                              scale=None, clip_wh=meta["clip_wh"])
    _write(ref, metas, rpn_conf, results_path)
    return dets.shape[0]


class _Pipelined:
    """Full batches go through one captured hipGraph per input shape (m3dssd_amd.pipeline.PipelinedDetector, refine=True):
    forward(batch k) || decode -> top-k -> NMS -> select -> m3d_refine_3d_ex of batch k-1; the host formats the text of batch
    k-1 while batch k runs."""

    def __init__(self, net, rpn_conf, results_path, dev):
        self.net, self.conf, self.path, self.dev = net, rpn_conf, results_path, dev
        self.pipes = {}
        self.open = None                      # (pipe, metas) of the batch in flight

    def submit(self, ims, metas):
        from ..pipeline import PipelinedDetector
        x = torch.cat([im if im.dim() == 4 else im[None] for im in ims]).to(self.dev, torch.float32)
        key = tuple(x.shape)
        pipe = self.pipes.get(key)
        if pipe is None:
            pipe = self.pipes[key] = PipelinedDetector(self.net, self.conf, key[0], key[2], key[3], refine=True)
        if self.open is not None and self.open[0] is not pipe:
            self.drain()
        res = pipe.step(x, meta=_meta_arrays(metas, self.conf))
        if res is not None:
            _write(res[2], self.open[1], self.conf, self.path)
        self.open = (pipe, metas)
        return len(metas)

    def drain(self):
        if self.open is not None:
            res = self.open[0].flush()
            _write(res[2], self.open[1], self.conf, self.path)
            self.open = None


def test_kitti_3d(dataset_test, net, rpn_conf, results_path, test_path, use_log=True, writer=None, phase="validation",
                  require_labels=True):
    """Same arguments as the reference (+ require_labels).  Returns (result text, stats dict) of get_official_eval_result.
    A missing label folder of `phase` raises FileNotFoundError like the reference does when it reads it
    (lib/rpn_util.py:1868-1876) -- a wrong test_path / phase must not look like a successful run without AP;
    require_labels=False writes the result files only and returns (None, None)."""
    os.makedirs(results_path, exist_ok=True)
    net = unwrap(net)                                      # scripts/test_rpn_3d.py:50-59 passes the nn.DataParallel wrapper
    dev = next(net.parameters()).device
    net.eval()
    bs = max(1, int(getattr(rpn_conf, "batch_size", 1)))
    ims, metas, n_done = [], [], 0
    pipe = _Pipelined(net, rpn_conf, results_path, dev)
    with torch.no_grad():
        for batch in dataset_test:
            im, meta = _unpack(batch, rpn_conf)
            im = torch.as_tensor(im)
            if ims and tuple(im.shape[-2:]) != tuple(ims[0].shape[-2:]):      # a different padded size: finish the open batch
                pipe.drain()
                n_done += _flush(ims, metas, net, rpn_conf, results_path, dev)
                ims, metas = [], []
            ims.append(im)
            metas.append(meta)
            if len(ims) == bs:
                n_done += pipe.submit(ims, metas)
                ims, metas = [], []
        pipe.drain()
        if ims:
            n_done += _flush(ims, metas, net, rpn_conf, results_path, dev)
    key = "datasets_{}".format(phase)
    sub = {"validation": "validation", "train": "training", "test": "testing"}[phase]
    gt_path = None
    if key in rpn_conf and rpn_conf[key]:
        gt_path = os.path.join(test_path, _get(rpn_conf[key][0], "name"), sub, "label_2")
    if gt_path is None or not os.path.isdir(gt_path):
        if require_labels:
            raise FileNotFoundError("test_kitti_3d: label folder of phase %r not found (%s); result files are in %s"
                                    % (phase, gt_path, results_path))
        logging.warning("test_kitti_3d: no label folder for phase %r (%s): result files written, no AP", phase, gt_path)
        return None, None
    dt_annos = get_label_annos(results_path)
    gt_annos = get_label_annos(gt_path)
    res, res_stats = get_official_eval_result(gt_annos, dt_annos, [0, 1, 2])
    if writer is not None:                                 # lib/rpn_util.py:1767-1768,1880-1896
        # the reference's step: the results folder is named results_<iteration> (file_parts(...)[1] without 'results_')
        test_iter = os.path.basename(os.path.normpath(results_path.replace("/data", ""))).replace("results_", "")
        step = int(test_iter) if test_iter.isdigit() else 0
        for lbl in rpn_conf.lbls:
            for item in ("aos", "3d", "bev", "image"):
                k = "{}_{}".format(lbl, item)
                if item == "3d" and lbl == "Car" and k + "_easy_R40" in res_stats:
                    writer.add_scalars("Test/" + k, {"easyR40": res_stats[k + "_easy_R40"], "modR40": res_stats[k + "_moderate_R40"],
                                                     "hardR40": res_stats[k + "_hard_R40"]}, step)
                if k + "_easy" in res_stats:
                    writer.add_scalars("Test/" + k, {"easy": res_stats[k + "_easy"], "mod": res_stats[k + "_moderate"],
                                                     "hard": res_stats[k + "_hard"]}, step)
    if use_log:
        logging.info(res)
    else:
        print(res)
    return res, res_stats
