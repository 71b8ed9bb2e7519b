"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's KITTI AP evaluator.

    lib/eval/rotate_iou.py:12-262      rotated-box intersection (the numba.cuda device functions: corners, point-in-
                                       quadrilateral, segment intersection, angular vertex sort, fan triangulation)
    lib/eval/eval.py:8-27              get_thresholds
    lib/eval/eval.py:30-81             clean_data (class / difficulty filtering)
    lib/eval/eval.py:84-149            image_box_overlap, d3_box_overlap_kernel
    lib/eval/eval.py:152-272           compute_statistics_jit (greedy GT <-> detection matching)
    lib/eval/eval.py:287-333           fused_compute_statistics
    lib/eval/eval.py:336-417           calculate_iou_partly
    lib/eval/eval.py:420-550           _prepare_data, eval_class
    lib/eval/eval.py:553-614           get_mAP, get_mAP_R40, do_eval
    lib/eval/eval.py:638-747           get_official_eval_result (text + dict)
    lib/eval/kitti_common.py:293-345   get_label_anno / get_label_annos

Plain Python / numpy loops, no numba.  The rotated intersection follows numba's typing of the reference kernels: box
corners, segment intersections and the vertex sort in float32, the triangle fan summed in float64 (`/ 2.0` promotes), the
ratio stored as float32.  cos / sin / sqrt are numpy's float32 functions (the reference runs libdevice's: the last ulp may
differ, which moves an IoU by ~1e-7).  Pinned by tests/golden/kitti_eval.npz (tools/gen_golden_eval.py runs the reference's
own lib/eval code, numba stubbed out, on seeded synthetic labels / detections).
"""
import io
import math
import pathlib
import re

import numpy as np

F = np.float32


# ----------------------------------------------------------------------------------------------- rotated IoU
def _triangle_area(a, b, c):
    return float(F(F(F(a[0] - c[0]) * F(b[1] - c[1])) - F(F(a[1] - c[1]) * F(b[0] - c[0])))) / 2.0


def _area(pts, n):
    s = 0.0
    for i in range(n - 2):
        s += abs(_triangle_area(pts[0:2], pts[2 * i + 2:2 * i + 4], pts[2 * i + 4:2 * i + 6]))
    return s


def _sort_vertices(pts, n):
    if n <= 0:
        return
    cx, cy = F(0), F(0)
    for i in range(n):
        cx = F(cx + pts[2 * i])
        cy = F(cy + pts[2 * i + 1])
    cx, cy = F(float(cx) / n), F(float(cy) / n)
    vs = np.zeros(16, dtype=F)
    for i in range(n):
        v0, v1 = F(pts[2 * i] - cx), F(pts[2 * i + 1] - cy)
        d = F(np.sqrt(F(F(v0 * v0) + F(v1 * v1))))
        v0, v1 = F(v0 / d), F(v1 / d)
        if v1 < 0:
            v0 = F(-2.0 - float(v0))
        vs[i] = v0
    for i in range(1, n):
        if vs[i - 1] > vs[i]:
            temp, tx, ty = vs[i], pts[2 * i], pts[2 * i + 1]
            j = i
            while j > 0 and vs[j - 1] > temp:
                vs[j] = vs[j - 1]
                pts[j * 2] = pts[j * 2 - 2]
                pts[j * 2 + 1] = pts[j * 2 - 1]
                j -= 1
            vs[j] = temp
            pts[j * 2] = tx
            pts[j * 2 + 1] = ty


def _segment_intersection(p1, p2, i, j):
    A0, A1 = p1[2 * i], p1[2 * i + 1]
    B0, B1 = p1[2 * ((i + 1) % 4)], p1[2 * ((i + 1) % 4) + 1]
    C0, C1 = p2[2 * j], p2[2 * j + 1]
    D0, D1 = p2[2 * ((j + 1) % 4)], p2[2 * ((j + 1) % 4) + 1]
    BA0, BA1 = F(B0 - A0), F(B1 - A1)
    DA0, CA0 = F(D0 - A0), F(C0 - A0)
    DA1, CA1 = F(D1 - A1), F(C1 - A1)
    acd = F(DA1 * CA0) > F(CA1 * DA0)
    bcd = F(F(D1 - B1) * F(C0 - B0)) > F(F(C1 - B1) * F(D0 - B0))
    if acd != bcd:
        abc = F(CA1 * BA0) > F(BA1 * CA0)
        abd = F(DA1 * BA0) > F(BA1 * DA0)
        if abc != abd:
            DC0, DC1 = F(D0 - C0), F(D1 - C1)
            ABBA = F(F(A0 * B1) - F(B0 * A1))
            CDDC = F(F(C0 * D1) - F(D0 * C1))
            DH = F(F(BA1 * DC0) - F(BA0 * DC1))
            Dx = F(F(ABBA * DC0) - F(BA0 * CDDC))
            Dy = F(F(ABBA * DC1) - F(BA1 * CDDC))
            return F(Dx / DH), F(Dy / DH)
    return None


def _point_in_quad(px, py, c):
    ab0, ab1 = F(c[2] - c[0]), F(c[3] - c[1])
    ad0, ad1 = F(c[6] - c[0]), F(c[7] - c[1])
    ap0, ap1 = F(px - c[0]), F(py - c[1])
    abab = F(F(ab0 * ab0) + F(ab1 * ab1))
    abap = F(F(ab0 * ap0) + F(ab1 * ap1))
    adad = F(F(ad0 * ad0) + F(ad1 * ad1))
    adap = F(F(ad0 * ap0) + F(ad1 * ap1))
    return abab >= abap and abap >= 0 and adad >= adap and adap >= 0


def _corners(rb):
    ang = F(rb[4])
    ac, as_ = F(np.cos(ang)), F(np.sin(ang))
    cx, cy, xd, yd = F(rb[0]), F(rb[1]), F(rb[2]), F(rb[3])
    xs = [F(-float(xd) / 2), F(-float(xd) / 2), F(float(xd) / 2), F(float(xd) / 2)]
    ys = [F(-float(yd) / 2), F(float(yd) / 2), F(float(yd) / 2), F(-float(yd) / 2)]
    out = np.zeros(8, dtype=F)
    for i in range(4):
        out[2 * i] = F(F(F(ac * xs[i]) + F(as_ * ys[i])) + cx)
        out[2 * i + 1] = F(F(F(-as_ * xs[i]) + F(ac * ys[i])) + cy)
    return out


def _inter(rb1, rb2):
    c1, c2 = _corners(rb1), _corners(rb2)
    pts = np.zeros(16, dtype=F)
    n = 0
    for i in range(4):
        if _point_in_quad(c1[2 * i], c1[2 * i + 1], c2):
            pts[2 * n], pts[2 * n + 1] = c1[2 * i], c1[2 * i + 1]
            n += 1
        if _point_in_quad(c2[2 * i], c2[2 * i + 1], c1):
            pts[2 * n], pts[2 * n + 1] = c2[2 * i], c2[2 * i + 1]
            n += 1
    for i in range(4):
        for j in range(4):
            r = _segment_intersection(c1, c2, i, j)
            if r is not None:
                if n >= 8:          # the reference's 16-float scratch: more points cannot occur for two convex quadrilaterals
                    continue
                pts[2 * n], pts[2 * n + 1] = r
                n += 1
    _sort_vertices(pts, n)
    return _area(pts, n)


def rotate_iou_eval(boxes, query_boxes, criterion=-1):
    """rotate_iou_gpu_eval, rotate_iou.py:264-326: iou[n, k] = devRotateIoUEval(query_boxes[k], boxes[n], criterion)."""
    boxes = np.asarray(boxes).astype(F)
    query_boxes = np.asarray(query_boxes).astype(F)
    N, K = boxes.shape[0], query_boxes.shape[0]
    iou = np.zeros((N, K), dtype=F)
    for n in range(N):
        for k in range(K):
            r1, r2 = query_boxes[k], boxes[n]
            a1, a2 = F(r1[2] * r1[3]), F(r2[2] * r2[3])
            ai = _inter(r1, r2)
            if criterion == -1:
                v = ai / (float(F(a1 + a2)) - ai)
            elif criterion == 0:
                v = ai / float(a1)
            elif criterion == 1:
                v = ai / float(a2)
            else:
                v = ai
            iou[n, k] = F(v)
    return iou


# ----------------------------------------------------------------------------------------------- eval.py
def get_thresholds(scores, num_gt, num_sample_pts=41):
    scores = np.sort(np.asarray(scores, dtype=np.float64))[::-1]
    current_recall = 0
    thresholds = []
    for i, score in enumerate(scores):
        l_recall = (i + 1) / num_gt
        r_recall = (i + 2) / num_gt if i < (len(scores) - 1) else l_recall
        if ((r_recall - current_recall) < (current_recall - l_recall)) and (i < (len(scores) - 1)):
            continue
        thresholds.append(score)
        current_recall += 1 / (num_sample_pts - 1.0)
    return thresholds


CLASS_NAMES = ['car', 'pedestrian', 'cyclist', 'van', 'person_sitting', 'truck']
MIN_HEIGHT = [40, 25, 25]
MAX_OCCLUSION = [0, 1, 2]
MAX_TRUNCATION = [0.15, 0.3, 0.5]


def clean_data(gt_anno, dt_anno, current_class, difficulty):
    dc_bboxes, ignored_gt, ignored_dt = [], [], []
    cur = CLASS_NAMES[current_class].lower()
    num_valid_gt = 0
    for i in range(len(gt_anno["name"])):
        bbox = gt_anno["bbox"][i]
        name = gt_anno["name"][i].lower()
        height = bbox[3] - bbox[1]
        if name == cur:
            valid = 1
        elif cur == "pedestrian" and name == "person_sitting":
            valid = 0
        elif cur == "car" and name == "van":
            valid = 0
        else:
            valid = -1
        ignore = ((gt_anno["occluded"][i] > MAX_OCCLUSION[difficulty]) or (gt_anno["truncated"][i] > MAX_TRUNCATION[difficulty])
                  or (height <= MIN_HEIGHT[difficulty]))
        if valid == 1 and not ignore:
            ignored_gt.append(0)
            num_valid_gt += 1
        elif valid == 0 or (ignore and valid == 1):
            ignored_gt.append(1)
        else:
            ignored_gt.append(-1)
        if gt_anno["name"][i] == "DontCare":
            dc_bboxes.append(gt_anno["bbox"][i])
    for i in range(len(dt_anno["name"])):
        valid = 1 if dt_anno["name"][i].lower() == cur else -1
        height = abs(dt_anno["bbox"][i, 3] - dt_anno["bbox"][i, 1])
        if height < MIN_HEIGHT[difficulty]:
            ignored_dt.append(1)
        elif valid == 1:
            ignored_dt.append(0)
        else:
            ignored_dt.append(-1)
    return num_valid_gt, ignored_gt, ignored_dt, dc_bboxes


def image_box_overlap(boxes, query_boxes, criterion=-1):
    N, K = boxes.shape[0], query_boxes.shape[0]
    ov = np.zeros((N, K), dtype=boxes.dtype)
    for k in range(K):
        qa = (query_boxes[k, 2] - query_boxes[k, 0]) * (query_boxes[k, 3] - query_boxes[k, 1])
        for n in range(N):
            iw = min(boxes[n, 2], query_boxes[k, 2]) - max(boxes[n, 0], query_boxes[k, 0])
            if iw > 0:
                ih = min(boxes[n, 3], query_boxes[k, 3]) - max(boxes[n, 1], query_boxes[k, 1])
                if ih > 0:
                    if criterion == -1:
                        ua = (boxes[n, 2] - boxes[n, 0]) * (boxes[n, 3] - boxes[n, 1]) + qa - iw * ih
                    elif criterion == 0:
                        ua = (boxes[n, 2] - boxes[n, 0]) * (boxes[n, 3] - boxes[n, 1])
                    elif criterion == 1:
                        ua = qa
                    else:
                        ua = 1.0
                    ov[n, k] = iw * ih / ua
    return ov


def d3_box_overlap(boxes, qboxes, criterion=-1):
    rinc = rotate_iou_eval(boxes[:, [0, 2, 3, 5, 6]], qboxes[:, [0, 2, 3, 5, 6]], 2).astype(boxes.dtype)
    N, K = boxes.shape[0], qboxes.shape[0]
    for i in range(N):
        for j in range(K):
            if rinc[i, j] > 0:
                iw = min(boxes[i, 1], qboxes[j, 1]) - max(boxes[i, 1] - boxes[i, 4], qboxes[j, 1] - qboxes[j, 4])
                if iw > 0:
                    a1 = boxes[i, 3] * boxes[i, 4] * boxes[i, 5]
                    a2 = qboxes[j, 3] * qboxes[j, 4] * qboxes[j, 5]
                    inc = iw * rinc[i, j]
                    ua = (a1 + a2 - inc) if criterion == -1 else (a1 if criterion == 0 else (a2 if criterion == 1 else inc))
                    rinc[i, j] = inc / ua
                else:
                    rinc[i, j] = 0.0
    return rinc


def compute_statistics(overlaps, gt_datas, dt_datas, ignored_gt, ignored_det, dc_bboxes, metric, min_overlap, thresh=0,
                       compute_fp=False, compute_aos=False):
    det_size, gt_size = dt_datas.shape[0], gt_datas.shape[0]
    dt_scores, dt_alphas, gt_alphas = dt_datas[:, -1], dt_datas[:, 4], gt_datas[:, 4]
    dt_bboxes = dt_datas[:, :4]
    assigned = [False] * det_size
    ign_thr = [False] * det_size
    if compute_fp:
        for i in range(det_size):
            if dt_scores[i] < thresh:
                ign_thr[i] = True
    NO_DET = -10000000
    tp, fp, fn, similarity = 0, 0, 0, 0
    thresholds, delta = [], []
    for i in range(gt_size):
        if ignored_gt[i] == -1:
            continue
        det_idx, valid_detection, max_overlap, assigned_ignored = -1, NO_DET, 0, False
        for j in range(det_size):
            if ignored_det[j] == -1 or assigned[j] or ign_thr[j]:
                continue
            overlap, sc = overlaps[j, i], dt_scores[j]
            if (not compute_fp) and overlap > min_overlap and sc > valid_detection:
                det_idx, valid_detection = j, sc
            elif compute_fp and overlap > min_overlap and (overlap > max_overlap or assigned_ignored) and ignored_det[j] == 0:
                max_overlap, det_idx, valid_detection, assigned_ignored = overlap, j, 1, False
            elif compute_fp and overlap > min_overlap and valid_detection == NO_DET and ignored_det[j] == 1:
                det_idx, valid_detection, assigned_ignored = j, 1, True
        if valid_detection == NO_DET and ignored_gt[i] == 0:
            fn += 1
        elif valid_detection != NO_DET and (ignored_gt[i] == 1 or ignored_det[det_idx] == 1):
            assigned[det_idx] = True
        elif valid_detection != NO_DET:
            tp += 1
            thresholds.append(dt_scores[det_idx])
            if compute_aos:
                delta.append(gt_alphas[i] - dt_alphas[det_idx])
            assigned[det_idx] = True
    if compute_fp:
        for i in range(det_size):
            if not (assigned[i] or ignored_det[i] == -1 or ignored_det[i] == 1 or ign_thr[i]):
                fp += 1
        nstuff = 0
        if metric == 0:
            ov_dc = image_box_overlap(dt_bboxes, dc_bboxes, 0)
            for i in range(dc_bboxes.shape[0]):
                for j in range(det_size):
                    if assigned[j] or ignored_det[j] == -1 or ignored_det[j] == 1 or ign_thr[j]:
                        continue
                    if ov_dc[j, i] > min_overlap:
                        assigned[j] = True
                        nstuff += 1
        fp -= nstuff
        if compute_aos:
            tmp = np.zeros((fp + len(delta),))
            for i in range(len(delta)):
                tmp[i + fp] = (1.0 + np.cos(delta[i])) / 2.0
            similarity = np.sum(tmp) if (tp > 0 or fp > 0) else -1
    return tp, fp, fn, similarity, np.asarray(thresholds, dtype=np.float64)


def get_split_parts(num, num_part):
    same, rem = num // num_part, num % num_part
    if same == 0:
        return [num]
    return [same] * num_part + ([rem] if rem else [])


def _boxes_for(annos, metric):
    if metric == 0:
        return np.concatenate([a["bbox"] for a in annos], 0)
    if metric == 1:
        loc = np.concatenate([a["location"][:, [0, 2]] for a in annos], 0)
        dims = np.concatenate([a["dimensions"][:, [0, 2]] for a in annos], 0)
    else:
        loc = np.concatenate([a["location"] for a in annos], 0)
        dims = np.concatenate([a["dimensions"] for a in annos], 0)
    rots = np.concatenate([a["rotation_y"] for a in annos], 0)
    return np.concatenate([loc, dims, rots[..., np.newaxis]], axis=1)


def calculate_iou_partly(gt_annos, dt_annos, metric, num_parts=50):
    assert len(gt_annos) == len(dt_annos)
    total_dt_num = np.stack([len(a["name"]) for a in dt_annos], 0)
    total_gt_num = np.stack([len(a["name"]) for a in gt_annos], 0)
    split_parts = get_split_parts(len(gt_annos), num_parts)
    parted, idx = [], 0
    for num_part in split_parts:
        g, d = _boxes_for(gt_annos[idx:idx + num_part], metric), _boxes_for(dt_annos[idx:idx + num_part], metric)
        if metric == 0:
            part = image_box_overlap(g, d)
        elif metric == 1:
            part = rotate_iou_eval(g, d, -1).astype(g.dtype).astype(np.float64)
        elif metric == 2:
            part = d3_box_overlap(g, d).astype(np.float64)
        else:
            raise ValueError("unknown metric")
        parted.append(part)
        idx += num_part
    overlaps, idx = [], 0
    for j, num_part in enumerate(split_parts):
        gi, di = 0, 0
        for i in range(num_part):
            gn, dn = total_gt_num[idx + i], total_dt_num[idx + i]
            overlaps.append(parted[j][gi:gi + gn, di:di + dn])
            gi += gn
            di += dn
        idx += num_part
    return overlaps, parted, total_gt_num, total_dt_num


def _prepare_data(gt_annos, dt_annos, current_class, difficulty):
    gt_datas_list, dt_datas_list, total_dc_num = [], [], []
    ignored_gts, ignored_dets, dontcares = [], [], []
    total_num_valid_gt = 0
    for i in range(len(gt_annos)):
        num_valid_gt, ignored_gt, ignored_det, dc = clean_data(gt_annos[i], dt_annos[i], current_class, difficulty)
        ignored_gts.append(np.array(ignored_gt, dtype=np.int64))
        ignored_dets.append(np.array(ignored_det, dtype=np.int64))
        dc = np.zeros((0, 4)).astype(np.float64) if len(dc) == 0 else np.stack(dc, 0).astype(np.float64)
        total_dc_num.append(dc.shape[0])
        dontcares.append(dc)
        total_num_valid_gt += num_valid_gt
        gt_datas_list.append(np.concatenate([gt_annos[i]["bbox"], gt_annos[i]["alpha"][..., np.newaxis]], 1))
        dt_datas_list.append(np.concatenate([dt_annos[i]["bbox"], dt_annos[i]["alpha"][..., np.newaxis],
                                             dt_annos[i]["score"][..., np.newaxis]], 1))
    return (gt_datas_list, dt_datas_list, ignored_gts, ignored_dets, dontcares, np.stack(total_dc_num, axis=0),
            total_num_valid_gt)


def eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False, num_parts=50):
    assert len(gt_annos) == len(dt_annos)
    split_parts = get_split_parts(len(gt_annos), num_parts)
    overlaps, parted_overlaps, total_dt_num, total_gt_num = calculate_iou_partly(dt_annos, gt_annos, metric, num_parts)
    N_SAMPLE_PTS = 41
    shape = [len(current_classes), len(difficultys), len(min_overlaps), N_SAMPLE_PTS]
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for m, current_class in enumerate(current_classes):
        for l, difficulty in enumerate(difficultys):
            (gt_datas_list, dt_datas_list, ignored_gts, ignored_dets, dontcares, total_dc_num,
             total_num_valid_gt) = _prepare_data(gt_annos, dt_annos, current_class, difficulty)
            for k, min_overlap in enumerate(min_overlaps[:, metric, m]):
                thresholdss = []
                for i in range(len(gt_annos)):
                    r = compute_statistics(overlaps[i], gt_datas_list[i], dt_datas_list[i], ignored_gts[i], ignored_dets[i],
                                           dontcares[i], metric, min_overlap=min_overlap, thresh=0.0, compute_fp=False)
                    thresholdss += r[4].tolist()
                thresholds = np.array(get_thresholds(np.array(thresholdss), total_num_valid_gt))
                pr = np.zeros([len(thresholds), 4])
                # fused_compute_statistics over the parts == the per-image loop (the parts only batch the IoU computation)
                for i in range(len(gt_annos)):
                    for t, thresh in enumerate(thresholds):
                        tp, fp, fn, sim, _ = compute_statistics(overlaps[i], gt_datas_list[i], dt_datas_list[i], ignored_gts[i],
                                                                ignored_dets[i], dontcares[i], metric, min_overlap=min_overlap,
                                                                thresh=thresh, compute_fp=True, compute_aos=compute_aos)
                        pr[t, 0] += tp
                        pr[t, 1] += fp
                        pr[t, 2] += fn
                        if sim != -1:
                            pr[t, 3] += sim
                with np.errstate(invalid="ignore", divide="ignore"):
                    for i in range(len(thresholds)):
                        recall[m, l, k, i] = pr[i, 0] / (pr[i, 0] + pr[i, 2])
                        precision[m, l, k, i] = pr[i, 0] / (pr[i, 0] + pr[i, 1])
                        if compute_aos:
                            aos[m, l, k, i] = pr[i, 3] / (pr[i, 0] + pr[i, 1])
                for i in range(len(thresholds)):
                    precision[m, l, k, i] = np.max(precision[m, l, k, i:], axis=-1)
                    recall[m, l, k, i] = np.max(recall[m, l, k, i:], axis=-1)
                    if compute_aos:
                        aos[m, l, k, i] = np.max(aos[m, l, k, i:], axis=-1)
    return {"recall": recall, "precision": precision, "orientation": aos}


def get_mAP(prec):
    sums = 0
    for i in range(0, prec.shape[-1], 4):
        sums = sums + prec[..., i]
    return sums / 11 * 100


def get_mAP_R40(prec):
    sums = 0
    for i in range(1, prec.shape[-1]):
        sums = sums + prec[..., i]
    return sums / 40 * 100


def do_eval(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos=False):
    difficultys = [0, 1, 2]
    ret = eval_class(gt_annos, dt_annos, current_classes, difficultys, 0, min_overlaps, compute_aos)
    mAP_bbox, mAP_bbox_R40 = get_mAP(ret["precision"]), get_mAP_R40(ret["precision"])
    mAP_aos = mAP_aos_R40 = None
    if compute_aos:
        mAP_aos, mAP_aos_R40 = get_mAP(ret["orientation"]), get_mAP_R40(ret["orientation"])
    ret = eval_class(gt_annos, dt_annos, current_classes, difficultys, 1, min_overlaps)
    mAP_bev, mAP_bev_R40 = get_mAP(ret["precision"]), get_mAP_R40(ret["precision"])
    ret = eval_class(gt_annos, dt_annos, current_classes, difficultys, 2, min_overlaps)
    mAP_3d, mAP_3d_R40 = get_mAP(ret["precision"]), get_mAP_R40(ret["precision"])
    return mAP_bbox, mAP_bev, mAP_3d, mAP_aos, mAP_bbox_R40, mAP_bev_R40, mAP_3d_R40, mAP_aos_R40


CLASS_TO_NAME = {0: 'Car', 1: 'Pedestrian', 2: 'Cyclist', 3: 'Van', 4: 'Person_sitting', 5: 'Truck'}


def _pline(s):
    return s + "\n"


def get_official_eval_result(gt_annos, dt_annos, current_classes, do_eval_fn=None):
    """-> (result text, dict) exactly as lib/eval/eval.py:638-747 formats them."""
    overlap_0_7 = np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.7], [0.7, 0.5, 0.5, 0.7, 0.5, 0.7], [0.7, 0.5, 0.5, 0.7, 0.5, 0.7]])
    min_overlaps = overlap_0_7[np.newaxis, :, :]
    name_to_class = {v: n for n, v in CLASS_TO_NAME.items()}
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    current_classes = [name_to_class[c] if isinstance(c, str) else c for c in current_classes]
    min_overlaps = min_overlaps[:, :, current_classes]
    compute_aos = False
    for anno in dt_annos:
        if anno['alpha'].shape[0] != 0:
            if anno['alpha'][0] != -10:
                compute_aos = True
            break
    fn = do_eval if do_eval_fn is None else do_eval_fn
    mAPbbox, mAPbev, mAP3d, mAPaos, mAPbbox_R40, mAPbev_R40, mAP3d_R40, mAPaos_R40 = fn(
        gt_annos, dt_annos, current_classes, min_overlaps, compute_aos)
    result, ret = '', {}
    for j, curcls in enumerate(current_classes):
        nm = CLASS_TO_NAME[curcls]
        for i in range(min_overlaps.shape[0]):
            result += _pline(f"{nm} " "AP@{:.2f}, {:.2f}, {:.2f}:".format(*min_overlaps[i, :, j]))
            result += _pline(f"bbox AP:{mAPbbox[j, 0, i]:.4f}, {mAPbbox[j, 1, i]:.4f}, {mAPbbox[j, 2, i]:.4f}")
            result += _pline(f"bev  AP:{mAPbev[j, 0, i]:.4f}, {mAPbev[j, 1, i]:.4f}, {mAPbev[j, 2, i]:.4f}")
            result += _pline(f"3d   AP:{mAP3d[j, 0, i]:.4f}, {mAP3d[j, 1, i]:.4f}, {mAP3d[j, 2, i]:.4f}")
            if compute_aos:
                result += _pline(f"aos  AP:{mAPaos[j, 0, i]:.2f}, {mAPaos[j, 1, i]:.2f}, {mAPaos[j, 2, i]:.2f}")
                if i == 0:
                    for d, dn in enumerate(("easy", "moderate", "hard")):
                        ret['%s_aos_%s' % (nm, dn)] = mAPaos[j, d, 0]
            result += _pline(f"{nm} " "AP_R40@{:.2f}, {:.2f}, {:.2f}:".format(*min_overlaps[i, :, j]))
            result += _pline(f"bbox AP:{mAPbbox_R40[j, 0, i]:.4f}, {mAPbbox_R40[j, 1, i]:.4f}, {mAPbbox_R40[j, 2, i]:.4f}")
            result += _pline(f"bev  AP:{mAPbev_R40[j, 0, i]:.4f}, {mAPbev_R40[j, 1, i]:.4f}, {mAPbev_R40[j, 2, i]:.4f}")
            result += _pline(f"3d   AP:{mAP3d_R40[j, 0, i]:.4f}, {mAP3d_R40[j, 1, i]:.4f}, {mAP3d_R40[j, 2, i]:.4f}")
            if compute_aos:
                result += _pline(f"aos  AP:{mAPaos_R40[j, 0, i]:.2f}, {mAPaos_R40[j, 1, i]:.2f}, {mAPaos_R40[j, 2, i]:.2f}")
                if i == 0:
                    for d, dn in enumerate(("easy", "moderate", "hard")):
                        ret['%s_aos_%s_R40' % (nm, dn)] = mAPaos_R40[j, d, 0]
            if i == 0:
                for d, dn in enumerate(("easy", "moderate", "hard")):
                    ret['%s_3d_%s' % (nm, dn)] = mAP3d[j, d, 0]
                    ret['%s_bev_%s' % (nm, dn)] = mAPbev[j, d, 0]
                    ret['%s_image_%s' % (nm, dn)] = mAPbbox[j, d, 0]
                    ret['%s_3d_%s_R40' % (nm, dn)] = mAP3d_R40[j, d, 0]
                    ret['%s_bev_%s_R40' % (nm, dn)] = mAPbev_R40[j, d, 0]
                    ret['%s_image_%s_R40' % (nm, dn)] = mAPbbox_R40[j, d, 0]
    return result, ret


# ----------------------------------------------------------------------------------------------- kitti_common.py
def get_label_anno(label_path):
    with open(label_path, 'r') as f:
        lines = f.readlines()
    content = [line.strip().split(' ') for line in lines]
    a = {}
    a['name'] = np.array([x[0] for x in content])
    a['truncated'] = np.array([float(x[1]) for x in content])
    a['occluded'] = np.array([int(x[2]) for x in content])
    a['alpha'] = np.array([float(x[3]) for x in content])
    a['bbox'] = np.array([[float(v) for v in x[4:8]] for x in content]).reshape(-1, 4)
    a['dimensions'] = np.array([[float(v) for v in x[8:11]] for x in content]).reshape(-1, 3)[:, [2, 0, 1]]
    a['location'] = np.array([[float(v) for v in x[11:14]] for x in content]).reshape(-1, 3)
    a['rotation_y'] = np.array([float(x[14]) for x in content]).reshape(-1)
    if len(content) != 0 and len(content[0]) == 16:
        a['score'] = np.array([float(x[15]) for x in content])
    else:
        a['score'] = np.zeros([len(a['bbox'])])
    return a


def get_label_annos(label_folder, image_ids=None):
    if image_ids is None:
        prog = re.compile(r'^\d{6}.txt$')
        image_ids = sorted(int(p.stem) for p in pathlib.Path(label_folder).glob('*.txt') if prog.match(p.name))
    if not isinstance(image_ids, list):
        image_ids = list(range(image_ids))
    folder = pathlib.Path(label_folder)
    return [get_label_anno(folder / ("{:06d}".format(i) + '.txt')) for i in image_ids]
