#!/usr/bin/env python
"""Golden vectors for the KITTI AP evaluator from the REFERENCE's own lib/eval code.

Build-container only (needs /root/reference).  numba is not installed and the rotated IoU is a numba.cuda kernel, so:

* ``numba.jit`` / ``numba.cuda.jit`` are stubbed to identity decorators, ``numba.prange`` to ``range``, ``cuda.local.array`` to
  a zero-filled float32 numpy array: the reference's functions then run as plain Python on numpy scalars -- every device
  function of lib/eval/rotate_iou.py (rbbox_to_corners, point_in_quadrilateral, line_segment_intersection,
  sort_vertex_in_convex_polygon, area, inter, devRotateIoUEval) and every jitted function of lib/eval/eval.py
  (get_thresholds, image_box_overlap, d3_box_overlap_kernel, compute_statistics_jit, fused_compute_statistics);
* the kernel launcher ``rotate_iou_gpu_eval`` (grid / shared-memory plumbing that cannot run without CUDA) is replaced by a loop
  that calls the reference's own ``devRotateIoUEval(query_box[k], box[n], criterion)`` for every pair, which is what the kernel
  computes (rotate_iou.py:255-262);
* lib/eval/eval.py is loaded from its source with the package-relative import resolved to that patched module.

Inputs are seeded synthetic KITTI label / result files (cars, pedestrians, cyclists, vans, DontCare regions; detections =
jittered ground truth + false positives + misses) written to a temporary directory and parsed by the reference's
``kitti_common.get_label_annos``.  The label / result TEXT is stored in the fixture so the tests re-create the files.
Writes tests/golden/kitti_eval.npz (data only).
"""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")


def _identity_jit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


def _load_reference_eval():
    numba = types.ModuleType("numba")
    numba.jit = _identity_jit
    numba.prange = range
    numba.float32 = np.float32
    cuda = types.ModuleType("numba.cuda")
    cuda.jit = _identity_jit

    class _Local:
        @staticmethod
        def array(shape, dtype=None):
            n = shape[0] if isinstance(shape, tuple) else shape
            return np.zeros(2 * n, dtype=np.float32)      # room to spare: the device code never bounds-checks its scratch
    cuda.local = _Local
    numba.cuda = cuda
    sys.modules["numba"], sys.modules["numba.cuda"] = numba, cuda
    from unittest.mock import MagicMock
    for name in ("skimage", "skimage.io"):          # imported by kitti_common for image helpers the label reader never calls
        sys.modules.setdefault(name, MagicMock())
    pkg = types.ModuleType("refeval")
    pkg.__path__ = [os.path.join(REF, "lib", "eval")]
    sys.modules["refeval"] = pkg

    def load(name):
        spec = importlib.util.spec_from_file_location("refeval." + name, os.path.join(REF, "lib", "eval", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["refeval." + name] = mod
        spec.loader.exec_module(mod)
        return mod

    riou = load("rotate_iou")

    def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
        boxes32, q32 = boxes.astype(np.float32), query_boxes.astype(np.float32)
        N, K = boxes32.shape[0], q32.shape[0]
        iou = np.zeros((N, K), dtype=np.float32)
        with np.errstate(all="ignore"):
            for n in range(N):
                for k in range(K):
                    iou[n, k] = riou.devRotateIoUEval(q32[k], boxes32[n], criterion)
        return iou.astype(boxes.dtype)
    riou.rotate_iou_gpu_eval = rotate_iou_gpu_eval
    ev = load("eval")
    kc = load("kitti_common")
    return riou, ev, kc


def synth_labels(rng, n_images):
    """-> (gt_texts, dt_texts): KITTI label_2 / result file contents per image."""
    names = ["Car", "Pedestrian", "Cyclist", "Van", "Person_sitting", "DontCare", "Truck"]
    size = {"Car": (1.5, 1.6, 3.9), "Van": (2.0, 1.9, 5.0), "Truck": (3.2, 2.6, 10.0), "Pedestrian": (1.75, 0.6, 0.8),
            "Person_sitting": (1.3, 0.6, 0.8), "Cyclist": (1.7, 0.6, 1.8)}
    f, cu, cv = 721.5, 609.6, 172.9
    gts, dts = [], []
    for _ in range(n_images):
        g_lines, d_lines = [], []
        for _ in range(int(rng.integers(3, 9))):
            nm = names[int(rng.choice(len(names), p=[0.4, 0.15, 0.12, 0.1, 0.05, 0.13, 0.05]))]
            if nm == "DontCare":
                x1, y1 = rng.uniform(0, 1100), rng.uniform(100, 250)
                g_lines.append("DontCare -1 -1 -10 %.2f %.2f %.2f %.2f -1 -1 -1 -1000 -1000 -1000 -10" % (
                    x1, y1, x1 + rng.uniform(30, 150), y1 + rng.uniform(20, 80)))
                continue
            h, w, l = (v * rng.uniform(0.9, 1.1) for v in size[nm])
            z = rng.uniform(6, 55)
            x = rng.uniform(-0.45, 0.45) * z
            y = rng.uniform(1.4, 1.9)
            ry = rng.uniform(-np.pi, np.pi)
            alpha = ry - np.arctan2(x, z)
            u, v = f * x / z + cu, f * (y - h / 2) / z + cv
            bw, bh = f * max(l, w) * rng.uniform(0.6, 1.0) / z, f * h / z
            box = (u - bw / 2, v - bh / 2, u + bw / 2, v + bh / 2)
            trunc = float(rng.choice([0.0, 0.0, 0.1, 0.25, 0.4, 0.6]))
            occ = int(rng.choice([0, 0, 1, 2, 3]))
            g_lines.append("%s %.2f %d %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f" % (
                nm, trunc, occ, alpha, *box, h, w, l, x, y, z, ry))
            if nm in ("Car", "Pedestrian", "Cyclist", "Van") and rng.uniform() < 0.8:      # a detection of this object
                j = rng.normal(0, 1, 12)
                dn = nm if rng.uniform() < 0.9 else "Car"
                sc = float(np.clip(rng.uniform(0.2, 1.0), 0, 1))
                d_lines.append("%s -1 -1 %.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f" % (
                    dn, alpha + 0.1 * j[0], box[0] + 2 * j[1], box[1] + 2 * j[2], box[2] + 2 * j[3], box[3] + 2 * j[4],
                    h * (1 + 0.03 * j[5]), w * (1 + 0.03 * j[6]), l * (1 + 0.03 * j[7]), x + 0.06 * j[8] * z / 20, y + 0.03 * j[9],
                    z + 0.12 * j[10] * z / 20, ry + 0.05 * j[11], sc))
        for _ in range(int(rng.integers(0, 4))):                                            # false positives
            nm = ["Car", "Pedestrian", "Cyclist"][int(rng.integers(0, 3))]
            h, w, l = size[nm]
            z = rng.uniform(6, 55)
            x = rng.uniform(-0.45, 0.45) * z
            u, v = f * x / z + cu, f * (1.65 - h / 2) / z + cv
            bw, bh = f * l * 0.8 / z, f * h / z
            d_lines.append("%s -1 -1 %.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f" % (
                nm, rng.uniform(-3, 3), u - bw / 2, v - bh / 2, u + bw / 2, v + bh / 2, h, w, l, x, 1.65, z, rng.uniform(-3, 3),
                rng.uniform(0.05, 0.9)))
        gts.append("\n".join(g_lines) + ("\n" if g_lines else ""))
        dts.append("\n".join(d_lines) + ("\n" if d_lines else ""))
    return gts, dts


def write_files(folder, texts):
    os.makedirs(folder, exist_ok=True)
    for i, t in enumerate(texts):
        with open(os.path.join(folder, "%06d.txt" % i), "w") as f:
            f.write(t)


def main():
    riou, ev, kc = _load_reference_eval()
    rng = np.random.default_rng(20260928)
    gts, dts = synth_labels(rng, 40)
    gts[5] = ""                       # an image without labels
    dts[7] = ""                       # an image without detections
    with tempfile.TemporaryDirectory() as tmp:
        write_files(os.path.join(tmp, "gt"), gts)
        write_files(os.path.join(tmp, "dt"), dts)
        gt_annos = kc.get_label_annos(os.path.join(tmp, "gt"))
        dt_annos = kc.get_label_annos(os.path.join(tmp, "dt"))
    with np.errstate(all="ignore"):
        text, stats = ev.get_official_eval_result(gt_annos, dt_annos, [0, 1, 2])
        # intermediate pins: a rotated-IoU matrix (all three criteria), a 3-D overlap matrix, one eval_class result
        rb = np.array([[0.0, 0.0, 4.0, 2.0, 0.3], [1.0, 0.5, 4.0, 2.0, -0.4], [10.0, 3.0, 1.0, 1.0, 1.0], [0.0, 0.0, 4.0, 2.0, 0.3],
                       [0.5, 0.1, 0.8, 0.6, 2.0], [2.0, 0.0, 4.0, 2.0, 0.3 + np.pi / 2]])
        qb = np.concatenate([rb[::-1] * np.array([1, 1, 1.1, 0.9, 1.0]), rng.uniform([-3, -3, 0.5, 0.5, -3], [3, 3, 5, 3, 3], (7, 5))])
        riou_m = {c: riou.rotate_iou_gpu_eval(rb, qb, c) for c in (-1, 0, 1, 2)}
        b7 = np.concatenate([rng.uniform([-5, 1, 5, 1, 1, 0.5, -3], [5, 2, 40, 5, 2, 2.5, 3], (9, 7))])
        q7 = b7[::-1] + rng.normal(0, 0.3, b7.shape) * np.array([1, 0.2, 1, 0.2, 0.1, 0.1, 0.3])
        d3 = ev.d3_box_overlap(b7, q7)
        mo = np.array([[0.7, 0.5, 0.5], [0.7, 0.5, 0.5], [0.7, 0.5, 0.5]])[np.newaxis]
        cls_ret = ev.eval_class(gt_annos, dt_annos, [0, 1, 2], [0, 1, 2], 1, mo, compute_aos=False)
    keys = sorted(stats)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "kitti_eval.npz")
    np.savez_compressed(path, gt_texts=np.array(gts), dt_texts=np.array(dts), result_text=text, stat_keys=np.array(keys),
                        stat_vals=np.array([stats[k] for k in keys], dtype=np.float64), rb=rb, qb=qb,
                        **{"riou_%d" % (c + 1): riou_m[c] for c in riou_m}, b7=b7, q7=q7, d3=d3,
                        bev_precision=cls_ret["precision"], bev_recall=cls_ret["recall"])
    print(path, "%.1f KB" % (os.path.getsize(path) / 1024))
    print(text)


if __name__ == "__main__":
    main()
