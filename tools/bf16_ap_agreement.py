#!/usr/bin/env python
"""A task-unit statement for the bf16 path WITHOUT KITTI (BASELINE.json configs[4]).

configs[4] is stated in AP3D on KITTI val; neither the dataset nor a trained checkpoint ships (`.gitignore:6` of the reference), so
AP3D parity is unpinned.  What CAN be measured in the same units: run the fp32 engine (the path that is within 1e-3 of the
reference) over N synthetic 1280x384 frames through the full result pipeline of `test_kitti_3d` (lib/rpn_util.py:1754-1893:
forward -> decode -> top-3000 -> NMS -> top-40 -> 3-D refinement -> KITTI result text), write those detections as PSEUDO GROUND
TRUTH label files, then score the bf16 engine's result files against them with the KITTI evaluator
(lib/eval/eval.py:638-747 `get_official_eval_result`, here m3dssd_amd.eval on the device).  fp32 scored against itself is 100 by
construction; the bf16 figure says how much of the fp32 detector's output survives the bf16 arithmetic at KITTI's own IoU
thresholds (Car 0.7 / Pedestrian, Cyclist 0.5; bbox, BEV, 3-D; AP_R11 and AP_R40).

This is NOT AP3D parity: the weights are random-init (synthetic recipe of SURVEY 8c), the "ground truth" is the fp32 detector
itself, every frame holds up to 40 pseudo objects.  It bounds the effect of the precision change in the units the task is judged in.

    python tools/bf16_ap_agreement.py [n_frames=64] [batch=8] [out_json]
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

P2 = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791], [0.0, 0.0, 1.0, 0.002745884],
               [0.0, 0.0, 0.0, 1.0]])                              # a KITTI calibration (training/calib/000000.txt, P2)


def _results(net, conf, frames, batch, out_dir, score_thresh):
    """The result files of test_kitti_3d for `frames` (host tensor [N, 3, H, W]) -> number of rows written."""
    from m3dssd_amd.host import refine as R
    from m3dssd_amd.host.detect import detect_batch
    os.makedirs(out_dir, exist_ok=True)
    dev = next(net.parameters()).device
    rows = 0
    for lo in range(0, frames.shape[0], batch):
        x = frames[lo:lo + batch].to(dev)
        dets, counts = detect_batch(net, x, conf)
        ref = R.refine_detections(dets, counts, P2, score_thresh=score_thresh,
                                  hill_climbing=bool(getattr(conf, "hill_climbing", True))).cpu().numpy()
        for b in range(x.shape[0]):
            text = R.kitti_text(ref[b], conf.lbls)
            rows += text.count("\n")
            with open(os.path.join(out_dir, "%06d.txt" % (lo + b)), "w") as f:
                f.write(text)
    return rows


GT_SHIFT_X, GT_SHIFT_Z = 1e-3, 5e-4        # metres


def _as_ground_truth(res_dir, gt_dir):
    """Result files -> label files: the score column dropped, truncated / occluded set to 0 (fully visible), so that KITTI's
    difficulty filters (height >= 40 / 25 / 25 px, occlusion, truncation: lib/eval/eval.py:55-116) act on the box height only.
    The 3-D location is shifted by (1 mm, 0.5 mm) in (x, z): the reference's rotated IoU (lib/eval/rotate_iou.py:223-253, float32
    edge tests) returns 0.0 for many rotated boxes against a BIT-IDENTICAL copy of themselves (pinned in tests/test_kitti_eval.py
    from the reference's own devRotateIoUEval), which would score the fp32 detector ~0 BEV / 3-D AP against itself; one
    millimetre apart the IoU is 0.999 and the fp32-vs-fp32 row reads 100."""
    os.makedirs(gt_dir, exist_ok=True)
    for name in sorted(os.listdir(res_dir)):
        out = []
        for line in open(os.path.join(res_dir, name)):
            f = line.split()
            f[11] = "%.6f" % (float(f[11]) + GT_SHIFT_X)
            f[13] = "%.6f" % (float(f[13]) + GT_SHIFT_Z)
            out.append(" ".join([f[0], "0.00", "0"] + f[3:15]))
        with open(os.path.join(gt_dir, name), "w") as g:
            g.write("\n".join(out) + ("\n" if out else ""))


def run(n_frames=64, batch=8, crop=(384, 1280), score_thresh=0.75, work_dir=None, seed=4321):
    from m3dssd_amd import synth
    from m3dssd_amd.eval import get_label_annos, get_official_eval_result
    from model.M3d_inference_align import build
    assert torch.cuda.is_available(), "needs the MI355X (no CPU fallback on the product path)"
    own = work_dir is None
    work_dir = work_dir or tempfile.mkdtemp(prefix="m3d_ap_")
    conf = synth.synth_conf(crop, 0, batch_size=batch, device="cuda:0")
    sd = synth.synth_state_dict(0)
    frames = torch.cat([synth.synth_frames(batch, crop, seed + i) for i in range(-(-n_frames // batch))])[:n_frames]
    frames[n_frames // 2:, :, :, (2 * crop[1]) // 3:] = 0.0           # half of the frames with the test-time zero border
    rows = {}
    for dt in ("f32", "bf16"):
        net = build(conf, "test")
        net.load_state_dict(sd, strict=True)
        net = net.to("cuda:0").set_compute_dtype(dt)
        with torch.no_grad():
            rows[dt] = _results(net, conf, frames, batch, os.path.join(work_dir, dt), score_thresh)
        del net
        torch.cuda.empty_cache()
    _as_ground_truth(os.path.join(work_dir, "f32"), os.path.join(work_dir, "gt"))
    gt = get_label_annos(os.path.join(work_dir, "gt"))
    out = {"frames": n_frames, "batch": batch, "crop": list(crop), "score_thresh": score_thresh,
           "rows_f32": rows["f32"], "rows_bf16": rows["bf16"],
           "note": "pseudo ground truth = the fp32 engine's own KITTI result rows on synthetic frames / weights; NOT AP3D parity"}
    for dt in ("f32", "bf16"):
        text, stats = get_official_eval_result(gt, get_label_annos(os.path.join(work_dir, dt)), [0, 1, 2])
        out[dt] = {k: round(float(v), 3) for k, v in stats.items()}
        out[dt + "_text"] = text
    names = np.concatenate([a["name"] for a in gt]) if gt else np.array([])
    out["gt_per_class"] = {c: int((names == c).sum()) for c in conf.lbls}
    if own:
        import shutil
        shutil.rmtree(work_dir, ignore_errors=True)
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    r = run(n, b)
    print(r.pop("f32_text"))
    print(r.pop("bf16_text"))
    print(json.dumps(r, indent=1))
    if len(sys.argv) > 3:
        json.dump(r, open(sys.argv[3], "w"), indent=1)
