#!/usr/bin/env python
"""Golden vectors for the post-NMS 3-D refinement: the reference's own ``hill_climb``,
``test_projection``, ``project_3d`` (lib/rpn_util.py:652-708,2015-2050,921-970) and ``convertAlpha2Rot`` / ``convertRot2Alpha``
(lib/util.py:516-535) run here on seeded detections with a KITTI-like projection matrix.  The per-box loop of test_kitti_3d
(lib/rpn_util.py:1801-1852) is inline code in the reference, not a function: it is re-run below statement by statement on the
np.float32 scalars `box = aboxes[boxind, :]` yields, with the reference's functions doing all the arithmetic.  The reference
pins numpy==1.18.1 (requirements.txt:50); this image has numpy 2.2, whose NEP-50 promotion would additionally round
`np.float32 + Python float` (convertAlpha2Rot's angle sum) to float32, which the reference's environment computes in float64:
that one argument is widened with float() before the call, everything else (np.float32 * np.float32 products, the float32
box_2d array) promotes identically under both numpy versions.  Writes tests/golden/refine.npz (data only)."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden  # noqa: E402


def main():
    gen_golden._install_stubs()
    import lib.rpn_util as R
    from lib.util import convertAlpha2Rot, convertRot2Alpha
    p2 = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791],
                   [0.0, 0.0, 1.0, 0.002745884], [0.0, 0.0, 0.0, 1.0]])          # a KITTI P2, padded to 4x4 as read_kitti_cal does
    p2_inv = np.linalg.inv(p2)
    rng = np.random.RandomState(11)
    rows, outs, texts = [], [], []
    lbls = ["Car", "Pedestrian", "Cyclist"]
    for i in range(48):
        z = float(rng.uniform(4, 60))
        xw, yw = float(rng.uniform(-0.4, 0.4) * z), float(rng.uniform(0.8, 2.0))
        w3d, h3d, l3d = float(rng.uniform(0.5, 2.0)), float(rng.uniform(1.3, 1.9)), float(rng.uniform(0.8, 4.5))
        ry = float(rng.uniform(-math.pi, math.pi))
        c2 = p2.dot(np.array([xw, yw - h3d / 2, z, 1.0]))
        x3d, y3d, z3d = float(c2[0] / c2[2]), float(c2[1] / c2[2]), float(c2[2])
        verts = R.project_3d(p2, xw, yw - h3d / 2, z, w3d, h3d, l3d, ry)
        jit = rng.uniform(-6, 6, size=4)                                          # an imperfect 2-D box, like a detector's
        x1, y1 = float(verts[:, 0].min() + jit[0]), float(verts[:, 1].min() + jit[1])
        x2, y2 = float(verts[:, 0].max() + jit[2]), float(verts[:, 1].max() + jit[3])
        alpha_in = float(convertRot2Alpha(ry + rng.uniform(-0.5, 0.5), z, xw))
        score = float(rng.uniform(0.5, 1.0))
        cls = int(rng.randint(1, 4))
        if i == 5:
            z3d = -2.0                                                            # behind the camera: hill_climb bails out
        row = np.array([x1, y1, x2, y2, score, cls, x3d, y3d, z3d, w3d, h3d, l3d, alpha_in, i], dtype=np.float32)
        rows.append(row)
        box = row                                                                 # np.float32 scalars, as in the reference loop
        x1, y1, x2, y2, score = box[0], box[1], box[2], box[3], box[4]
        width, height = (x2 - x1 + 1), (y2 - y1 + 1)
        x3d, y3d, z3d, w3d, h3d, l3d, ry3d = box[6], box[7], box[8], box[9], box[10], box[11], box[12]
        assert all(isinstance(v, np.float32) for v in (width, height, x3d * z3d, 1 * z3d))
        # ---- lib/rpn_util.py:1813-1847, reference functions, same statement order ----
        coord3d = np.linalg.inv(p2).dot(np.array([x3d * z3d, y3d * z3d, 1 * z3d, 1]))
        ry3d = convertAlpha2Rot(float(ry3d), coord3d[2], coord3d[0])              # numpy 1.18: np.float32 + float -> float64
        box_2d = np.array([x1, y1, width, height])
        assert box_2d.dtype == np.float32
        z3d, ry3d, verts_best = R.hill_climb(p2, p2_inv, box_2d, x3d, y3d, z3d, w3d, h3d, l3d, ry3d,
                                             step_r_init=0.3 * math.pi, r_lim=0.01)
        assert isinstance(z3d, np.float32) and not isinstance(ry3d, np.float32)
        coord3d = np.linalg.inv(p2).dot(np.array([x3d * z3d, y3d * z3d, 1 * z3d, 1]))
        alpha = convertRot2Alpha(ry3d, coord3d[2], coord3d[0])
        x3d, y3d, z3d = coord3d[0], coord3d[1], coord3d[2]
        y3d += h3d / 2
        vals = [float(v) for v in (alpha, x1, y1, x2, y2, h3d, w3d, l3d, x3d, y3d, z3d, ry3d, score)]
        cls = int(box[5])
        outs.append(vals)
        if score >= 0.75:
            texts.append(('{} -1 -1 {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} '
                          + '{:.6f} {:.6f}\n').format(lbls[cls - 1], *vals))
    ol, verts, b2, invalid = R.test_projection(p2, p2_inv, np.array([300.0, 150.0, 80.0, 60.0]), 340.0, 180.0, 20.0, 1.6, 1.5, 3.9, 0.3)
    path = os.path.join(gen_golden.OUT, "refine.npz")
    np.savez_compressed(path, p2=p2, rows=np.stack(rows), refined=np.asarray(outs, dtype=np.float64), text="".join(texts),
                        tp_ol=ol, tp_verts=verts, tp_invalid=invalid)
    print(path, "%.1f KB" % (os.path.getsize(path) / 1024), "lines", len(texts))


if __name__ == "__main__":
    main()
