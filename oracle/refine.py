"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's post-NMS 3-D refinement and KITTI result formatting
(SURVEY section 8f row 2).  The reference mixes np.float32 scalars (`box = aboxes[i]`) and Python floats; its environment pins
numpy==1.18.1 (requirements.txt:50, value-based promotion), under which np.float32 * np.float32 (x3d*z3d, y3d*z3d, the box
width / height, x + w - 1 inside test_projection) ROUNDS TO FLOAT32 while np.float32 + Python float is float64.  That is what
is restated here with explicit np.float32 arithmetic, so the result does not depend on the numpy installed (numpy >= 2 would
also round the angle sums to float32, which the reference's authors never ran).

    lib/util.py:516-535          convertAlpha2Rot / convertRot2Alpha
    lib/rpn_util.py:921-970      project_3d
    lib/rpn_util.py:2015-2050    test_projection   (score = -L1 distance between the 2-D box and the projected 3-D box)
    lib/rpn_util.py:652-708      hill_climb        (coordinate descent on depth / yaw with step halving)
    lib/rpn_util.py:1801-1852    per-box loop of test_kitti_3d and the result line format

Pinned by tests/golden/refine.npz (tools/gen_golden_refine.py runs the reference's own functions on seeded boxes).
"""
import math

import numpy as np

VERT_IDX = [0, 1, 2, 3, 4, 5, 6, 7, 0, 5, 4, 1, 2, 7, 6, 3]


def _wrap(a):
    while a > math.pi:
        a -= math.pi * 2
    while a < (-math.pi):
        a += math.pi * 2
    return a


def convert_alpha2rot(alpha, z3d, x3d):
    return _wrap(alpha + math.atan2(-z3d, x3d) + 0.5 * math.pi)


def convert_rot2alpha(ry3d, z3d, x3d):
    return _wrap(ry3d - math.atan2(-z3d, x3d) - 0.5 * math.pi)


def project_3d(p2, x3d, y3d, z3d, w3d, h3d, l3d, ry3d):
    """-> (verts [16, 2], corners_3d [3, 8])."""
    c, s = math.cos(ry3d), math.sin(ry3d)
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)
    xc = np.array([0, l3d, l3d, l3d, l3d, 0, 0, 0], dtype=np.float64) - l3d / 2
    yc = np.array([0, 0, h3d, h3d, 0, 0, h3d, h3d], dtype=np.float64) - h3d / 2
    zc = np.array([0, 0, 0, w3d, w3d, w3d, w3d, 0], dtype=np.float64) - w3d / 2
    corners = R.dot(np.array([xc, yc, zc])) + np.array([x3d, y3d, z3d], dtype=np.float64).reshape(3, 1)
    c2 = np.asarray(p2, dtype=np.float64).dot(np.vstack((corners, np.ones(8))))
    c2 = c2 / c2[2]
    return c2[:, VERT_IDX][:2].T.copy(), corners


def test_projection(p2, p2_inv, box_2d, cx, cy, z, w3d, h3d, l3d, rot_y):
    """-> (ol, verts, invalid)."""
    f32 = np.float32
    bx, by, bw, bh = (f32(v) for v in box_2d)                  # float32 array in the reference (np.array of np.float32 scalars)
    x, y = float(bx), float(by)
    x2, y2 = float(f32(f32(bx + bw) - f32(1))), float(f32(f32(by + bh) - f32(1)))      # np.float32 + np.float32 - 1
    if isinstance(z, np.float32):                              # cx, cy, z still the row's np.float32 scalars: float32 products
        v0, v1 = float(f32(f32(cx) * z)), float(f32(f32(cy) * z))
    else:                                                      # a stepped depth (np.float32 -+ Python float) is float64
        v0, v1 = float(cx) * float(z), float(cy) * float(z)
    c3 = np.asarray(p2_inv, dtype=np.float64).dot(np.array([v0, v1, float(z), 1], dtype=np.float64))
    w3d, h3d, l3d, rot_y = float(w3d), float(h3d), float(l3d), float(rot_y)
    verts, corners = project_3d(p2, c3[0], c3[1], c3[2], w3d, h3d, l3d, rot_y)
    invalid = bool(np.any(corners[2, :] <= 0))
    xn, yn, x2n, y2n = verts[:, 0].min(), verts[:, 1].min(), verts[:, 0].max(), verts[:, 1].max()
    ol = -(abs(x - xn) + abs(y - yn) + abs(x2 - x2n) + abs(y2 - y2n))
    return ol, verts, invalid


def hill_climb(p2, p2_inv, box_2d, x2d, y2d, z2d, w3d, h3d, l3d, ry3d, step_z_init=0, step_r_init=0, z_lim=0, r_lim=0,
               min_ol_dif=0.0):
    """-> (z2d, ry3d, verts_best)."""
    step_z, step_r = step_z_init, step_r_init
    ol_best, verts_best, invalid = test_projection(p2, p2_inv, box_2d, x2d, y2d, z2d, w3d, h3d, l3d, ry3d)
    if invalid:
        return z2d, ry3d, verts_best
    while step_z > z_lim or step_r > r_lim:
        if step_z > z_lim:
            ol_neg, v_neg, inv_neg = test_projection(p2, p2_inv, box_2d, x2d, y2d, float(z2d) - step_z, w3d, h3d, l3d, ry3d)
            ol_pos, v_pos, inv_pos = test_projection(p2, p2_inv, box_2d, x2d, y2d, float(z2d) + step_z, w3d, h3d, l3d, ry3d)
            if ((ol_pos - ol_best) <= min_ol_dif) and ((ol_neg - ol_best) <= min_ol_dif):
                step_z = step_z * 0.5
            elif (ol_pos - ol_best) > min_ol_dif and ol_pos > ol_neg and not inv_pos:
                z2d, ol_best, verts_best = float(z2d) + step_z, ol_pos, v_pos
            elif (ol_neg - ol_best) > min_ol_dif and not inv_neg:
                z2d, ol_best, verts_best = float(z2d) - step_z, ol_neg, v_neg
            else:
                step_z = step_z * 0.5
        if step_r > r_lim:
            ol_neg, v_neg, inv_neg = test_projection(p2, p2_inv, box_2d, x2d, y2d, z2d, w3d, h3d, l3d, ry3d - step_r)
            ol_pos, v_pos, inv_pos = test_projection(p2, p2_inv, box_2d, x2d, y2d, z2d, w3d, h3d, l3d, ry3d + step_r)
            if ((ol_pos - ol_best) <= min_ol_dif) and ((ol_neg - ol_best) <= min_ol_dif):
                step_r = step_r * 0.5
            elif (ol_pos - ol_best) > min_ol_dif and ol_pos > ol_neg and not inv_pos:
                ry3d, ol_best, verts_best = ry3d + step_r, ol_pos, v_pos
            elif (ol_neg - ol_best) > min_ol_dif and not inv_neg:
                ry3d, ol_best, verts_best = ry3d - step_r, ol_neg, v_neg
            else:
                step_r = step_r * 0.5
    return z2d, _wrap(ry3d), verts_best


def refine_row(row, p2, p2_inv, hill_climbing=True):
    """One aboxes row [x1, y1, x2, y2, score, cls, x3d, y3d, z3d, w3d, h3d, l3d, alpha, anchor] (rpn_util.py:1550) ->
    [alpha, x1, y1, x2, y2, h3d, w3d, l3d, x3d, y3d, z3d, ry3d, score] as written to the KITTI file (rpn_util.py:1813-1849)."""
    f32 = np.float32
    b = [f32(v) for v in row]                                      # np.float32 scalars, as `box = aboxes[boxind, :]` yields
    x1, y1, x2, y2, score = b[0], b[1], b[2], b[3], b[4]
    x3d, y3d, z3d, w3d, h3d, l3d = b[6], b[7], b[8], b[9], b[10], b[11]
    p2_inv = np.asarray(p2_inv, dtype=np.float64)

    def back_project(z):
        if isinstance(z, np.float32):                              # np.float32 * np.float32 -> float32, then widened
            v0, v1 = float(f32(x3d * z)), float(f32(y3d * z))
        else:
            v0, v1 = float(x3d) * float(z), float(y3d) * float(z)
        return p2_inv.dot(np.array([v0, v1, float(z), 1], dtype=np.float64))

    c3 = back_project(z3d)
    ry3d = convert_alpha2rot(float(b[12]), c3[2], c3[0])           # np.float32 + Python float -> float64 (numpy 1.18)
    if hill_climbing:
        box_2d = np.array([x1, y1, f32(f32(x2 - x1) + f32(1)), f32(f32(y2 - y1) + f32(1))], dtype=np.float32)
        z3d, ry3d, _ = hill_climb(p2, p2_inv, box_2d, x3d, y3d, z3d, w3d, h3d, l3d, ry3d, step_r_init=0.3 * math.pi, r_lim=0.01)
    c3 = back_project(z3d)
    alpha = convert_rot2alpha(ry3d, c3[2], c3[0])
    return [alpha, float(x1), float(y1), float(x2), float(y2), float(h3d), float(w3d), float(l3d), c3[0],
            c3[1] + float(h3d) / 2, c3[2], ry3d, float(score)]


def kitti_text(aboxes, p2, lbls, nms_topn_post=40, score_thresh=0.75, hill_climbing=True):
    """The text test_kitti_3d writes for one image (rpn_util.py:1801-1852)."""
    p2 = np.asarray(p2, dtype=np.float64)
    p2_inv = np.linalg.inv(p2)
    text = ""
    for i in range(min(nms_topn_post, aboxes.shape[0])):
        box = aboxes[i]
        if box[4] >= score_thresh:
            v = refine_row(box, p2, p2_inv, hill_climbing)
            text += ("{} -1 -1 {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} "
                     + "{:.6f} {:.6f}\n").format(lbls[int(box[5] - 1)], *v)
    return text
