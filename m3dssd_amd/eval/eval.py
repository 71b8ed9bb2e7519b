"""Official KITTI AP / AP_R40 tables (bbox, BEV, 3-D, AOS) with the reference's interface
(lib/eval/eval.py:638-747 ``get_official_eval_result(gt_annos, dt_annos, current_classes)`` -> (text, dict)).

MI355X-first layout of the work the reference spreads over numba-jitted loops and a numba.cuda kernel:

  * ALL rotated-IoU pairs of a metric are computed by ONE launch of ``m3d_rotate_iou_eval`` per part on the device (the
    reference's ``rotate_iou_gpu_eval``, lib/eval/rotate_iou.py:264-326); 2-D overlaps and the 3-D height intersection run
    in native host code (``m3d_eval_image_box_overlap``, ``m3d_eval_d3_overlap``);
  * the greedy GT <-> detection matching (``compute_statistics_jit`` / ``fused_compute_statistics``, eval.py:152-333) runs in
    native host code over whole parts (``m3d_eval_statistics`` / ``m3d_eval_fused_statistics``);
  * class / difficulty filtering, threshold sampling, the precision envelopes and the report text are numpy / Python.

There is no fallback: the natives live in libm3dssd_hip.so and the rotated IoU needs the ROCm device.
"""
import ctypes

import numpy as np
import torch

from .. import _hip

CLASS_TO_NAME = {0: "Car", 1: "Pedestrian", 2: "Cyclist", 3: "Van", 4: "Person_sitting", 5: "Truck"}
_CLASS_NAMES = ["car", "pedestrian", "cyclist", "van", "person_sitting", "truck"]
_MIN_HEIGHT = [40, 25, 25]
_MAX_OCCLUSION = [0, 1, 2]
_MAX_TRUNCATION = [0.15, 0.3, 0.5]
N_SAMPLE_PTS = 41


def _dp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


# ----------------------------------------------------------------------------------------------- overlaps
def rotate_iou_eval(boxes, query_boxes, criterion=-1, device=None):
    """[N,5] x [K,5] (centre, dims, angle) -> float32 [N,K] on the ROCm device (lib/eval/rotate_iou.py:264-326)."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    query_boxes = np.ascontiguousarray(query_boxes, dtype=np.float32)
    n, k = boxes.shape[0], query_boxes.shape[0]
    if n == 0 or k == 0:
        return np.zeros((n, k), dtype=np.float32)
    if not torch.cuda.is_available():
        raise NotImplementedError("rotate_iou_eval runs on the ROCm device (the reference's is a numba.cuda kernel); no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    b, q = torch.from_numpy(boxes).to(dev), torch.from_numpy(query_boxes).to(dev)
    out = torch.empty(n, k, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _hip.check(_hip.lib().m3d_rotate_iou_eval(b.data_ptr(), n, q.data_ptr(), k, int(criterion), out.data_ptr(), st))
    return out.cpu().numpy()


def image_box_overlap(boxes, query_boxes, criterion=-1):
    boxes, query_boxes = _c64(boxes), _c64(query_boxes)
    out = np.zeros((boxes.shape[0], query_boxes.shape[0]), dtype=np.float64)
    _hip.check(_hip.lib().m3d_eval_image_box_overlap(_dp(boxes), boxes.shape[0], _dp(query_boxes), query_boxes.shape[0],
                                                     int(criterion), _dp(out)))
    return out


def bev_box_overlap(boxes, qboxes, criterion=-1):
    return rotate_iou_eval(boxes, qboxes, criterion).astype(np.asarray(boxes).dtype)


def d3_box_overlap(boxes, qboxes, criterion=-1):
    boxes, qboxes = _c64(boxes), _c64(qboxes)
    rinc = _c64(rotate_iou_eval(boxes[:, [0, 2, 3, 5, 6]], qboxes[:, [0, 2, 3, 5, 6]], 2))
    _hip.check(_hip.lib().m3d_eval_d3_overlap(_dp(boxes), boxes.shape[0], _dp(qboxes), qboxes.shape[0], _dp(rinc), int(criterion)))
    return rinc


# ----------------------------------------------------------------------------------------------- filtering / thresholds
def clean_data(gt_anno, dt_anno, current_class, difficulty):
    """eval.py:30-81, vectorised: -> num_valid_gt, ignored_gt [int64], ignored_dt [int64], dc_bboxes [n,4]."""
    cur = _CLASS_NAMES[current_class]
    gname = np.char.lower(np.asarray(gt_anno["name"], dtype=str)) if len(gt_anno["name"]) else np.zeros(0, dtype=str)
    bbox = np.asarray(gt_anno["bbox"], dtype=np.float64).reshape(-1, 4)
    height = bbox[:, 3] - bbox[:, 1]
    valid = np.full(gname.shape[0], -1, dtype=np.int64)
    valid[gname == cur] = 1
    if cur == "pedestrian":
        valid[gname == "person_sitting"] = 0
    if cur == "car":
        valid[gname == "van"] = 0
    ignore = ((np.asarray(gt_anno["occluded"]) > _MAX_OCCLUSION[difficulty])
              | (np.asarray(gt_anno["truncated"]) > _MAX_TRUNCATION[difficulty]) | (height <= _MIN_HEIGHT[difficulty]))
    ignored_gt = np.full(gname.shape[0], -1, dtype=np.int64)
    ok = (valid == 1) & ~ignore
    ignored_gt[ok] = 0
    ignored_gt[~ok & ((valid == 0) | (ignore & (valid == 1)))] = 1
    dc = bbox[np.asarray(gt_anno["name"], dtype=str) == "DontCare"] if gname.shape[0] else np.zeros((0, 4))
    dname = np.char.lower(np.asarray(dt_anno["name"], dtype=str)) if len(dt_anno["name"]) else np.zeros(0, dtype=str)
    dbox = np.asarray(dt_anno["bbox"], dtype=np.float64).reshape(-1, 4)
    dh = np.abs(dbox[:, 3] - dbox[:, 1])
    ignored_dt = np.where(dh < _MIN_HEIGHT[difficulty], 1, np.where(dname == cur, 0, -1)).astype(np.int64)
    return int(ok.sum()), ignored_gt, ignored_dt, _c64(dc).reshape(-1, 4)


def get_thresholds(scores, num_gt, num_sample_pts=N_SAMPLE_PTS):
    scores = np.sort(np.asarray(scores, dtype=np.float64))[::-1]
    current_recall, thresholds, n = 0, [], len(scores)
    for i, score in enumerate(scores):
        l_recall = (i + 1) / num_gt
        r_recall = (i + 2) / num_gt if i < n - 1 else l_recall
        if (r_recall - current_recall) < (current_recall - l_recall) and i < n - 1:
            continue
        thresholds.append(score)
        current_recall += 1 / (num_sample_pts - 1.0)
    return thresholds


def get_split_parts(num, num_part):
    same, rem = num // num_part, num % num_part
    if same == 0:
        return [num]
    return [same] * num_part + ([rem] if rem else [])


def _boxes_for(annos, metric):
    if metric == 0:
        return np.concatenate([a["bbox"] for a in annos], 0)
    cols = [0, 2] if metric == 1 else [0, 1, 2]
    loc = np.concatenate([a["location"][:, cols] for a in annos], 0)
    dims = np.concatenate([a["dimensions"][:, cols] for a in annos], 0)
    rots = np.concatenate([a["rotation_y"] for a in annos], 0)
    return np.concatenate([loc, dims, rots[..., np.newaxis]], axis=1)


def calculate_iou_partly(gt_annos, dt_annos, metric, num_parts=50):
    """eval.py:336-417: the overlap matrix of every part in one shot ([sum boxes of arg 1][sum boxes of arg 2])."""
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
            part = bev_box_overlap(g, d).astype(np.float64)
        elif metric == 2:
            part = d3_box_overlap(g, d).astype(np.float64)
        else:
            raise ValueError("unknown metric")
        parted.append(np.ascontiguousarray(part))
        idx += num_part
    overlaps, idx = [], 0
    for j, num_part in enumerate(split_parts):
        gi = di = 0
        for i in range(num_part):
            gn, dn = total_gt_num[idx + i], total_dt_num[idx + i]
            overlaps.append(parted[j][gi:gi + gn, di:di + dn])
            gi += gn
            di += dn
        idx += num_part
    return overlaps, parted, total_gt_num, total_dt_num


def _prepare_data(gt_annos, dt_annos, current_class, difficulty):
    gt_datas_list, dt_datas_list, total_dc_num, ignored_gts, ignored_dets, dontcares = [], [], [], [], [], []
    total_num_valid_gt = 0
    for g, d in zip(gt_annos, dt_annos):
        nvalid, ig, idt, dc = clean_data(g, d, current_class, difficulty)
        ignored_gts.append(ig)
        ignored_dets.append(idt)
        total_dc_num.append(dc.shape[0])
        dontcares.append(dc)
        total_num_valid_gt += nvalid
        gt_datas_list.append(_c64(np.concatenate([g["bbox"], g["alpha"][..., np.newaxis]], 1)).reshape(-1, 5))
        dt_datas_list.append(_c64(np.concatenate([d["bbox"], d["alpha"][..., np.newaxis], d["score"][..., np.newaxis]], 1)).reshape(-1, 6))
    return gt_datas_list, dt_datas_list, ignored_gts, ignored_dets, dontcares, np.asarray(total_dc_num, dtype=np.int64), total_num_valid_gt


def eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False, num_parts=50):
    """eval.py:448-550 -> dict(recall, precision, orientation) of shape [class, difficulty, min_overlap, 41]."""
    assert len(gt_annos) == len(dt_annos), "len(gt_annos):{} ,len(dt_annos):{}".format(len(gt_annos), len(dt_annos))
    L = _hip.lib()
    split_parts = get_split_parts(len(gt_annos), num_parts)
    # the reference passes (dt, gt) here: overlaps are [detections][ground truths]
    overlaps, parted_overlaps, total_dt_num, total_gt_num = calculate_iou_partly(dt_annos, gt_annos, metric, num_parts)
    shape = [len(current_classes), len(difficultys), len(min_overlaps), N_SAMPLE_PTS]
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for m, current_class in enumerate(current_classes):
        for l, difficulty in enumerate(difficultys):
            (gt_datas_list, dt_datas_list, ignored_gts, ignored_dets, dontcares, total_dc_num,
             total_num_valid_gt) = _prepare_data(gt_annos, dt_annos, current_class, difficulty)
            for k, min_overlap in enumerate(min_overlaps[:, metric, m]):
                scores = []
                for i in range(len(gt_annos)):
                    ov = overlaps[i]
                    nd, ng = dt_datas_list[i].shape[0], gt_datas_list[i].shape[0]
                    thr = np.zeros(max(ng, 1), dtype=np.float64)
                    st = np.zeros(4, dtype=np.float64)
                    nthr = ctypes.c_int(0)
                    ovc = np.ascontiguousarray(ov, dtype=np.float64)
                    _hip.check(L.m3d_eval_statistics(_dp(ovc), max(ng, 1) if ovc.size == 0 else ovc.shape[1], _dp(gt_datas_list[i]), ng,
                                                     _dp(dt_datas_list[i]), nd, _dp(ignored_gts[i]), _dp(ignored_dets[i]),
                                                     _dp(dontcares[i]), dontcares[i].shape[0], metric, float(min_overlap), 0.0, 0, 0,
                                                     _dp(st), _dp(thr), ctypes.byref(nthr)))
                    scores.append(thr[:nthr.value])
                thresholds = np.array(get_thresholds(np.concatenate(scores) if scores else np.zeros(0), total_num_valid_gt),
                                      dtype=np.float64)
                pr = np.zeros([len(thresholds), 4], dtype=np.float64)
                idx = 0
                for j, num_part in enumerate(split_parts):
                    sl = slice(idx, idx + num_part)
                    gt_part = _c64(np.concatenate(gt_datas_list[sl], 0)).reshape(-1, 5)
                    dt_part = _c64(np.concatenate(dt_datas_list[sl], 0)).reshape(-1, 6)
                    dc_part = _c64(np.concatenate(dontcares[sl], 0)).reshape(-1, 4)
                    ig_part, id_part = _i64(np.concatenate(ignored_gts[sl], 0)), _i64(np.concatenate(ignored_dets[sl], 0))
                    po = parted_overlaps[j]
                    gn, dn, dcn = _i64(total_gt_num[sl]), _i64(total_dt_num[sl]), _i64(total_dc_num[sl])
                    _hip.check(L.m3d_eval_fused_statistics(_dp(po), po.shape[1] if po.ndim == 2 else 0, _dp(pr), _dp(gn), _dp(dn),
                                                           _dp(dcn), num_part, _dp(gt_part), _dp(dt_part), _dp(dc_part), _dp(ig_part),
                                                           _dp(id_part), metric, float(min_overlap), _dp(thresholds), len(thresholds),
                                                           1 if compute_aos else 0))
                    idx += num_part
                nt = len(thresholds)
                with np.errstate(invalid="ignore", divide="ignore"):
                    recall[m, l, k, :nt] = pr[:, 0] / (pr[:, 0] + pr[:, 2])
                    precision[m, l, k, :nt] = pr[:, 0] / (pr[:, 0] + pr[:, 1])
                    if compute_aos:
                        aos[m, l, k, :nt] = pr[:, 3] / (pr[:, 0] + pr[:, 1])
                # running maximum from the right over all 41 sample points (zeros past the last threshold take part, like the
                # reference's np.max(precision[m, l, k, i:]) for i < len(thresholds))
                for arr in (precision, recall) + ((aos,) if compute_aos else ()):
                    row = arr[m, l, k]
                    env = np.maximum.accumulate(row[::-1])[::-1]      # NaN entries propagate exactly like np.max does
                    row[:nt] = env[:nt]
    return {"recall": recall, "precision": precision, "orientation": aos}


def get_mAP(prec):
    return prec[..., 0::4].sum(-1) / 11 * 100


def get_mAP_R40(prec):
    return prec[..., 1:].sum(-1) / 40 * 100


def do_eval(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos=False, PR_detail_dict=None):
    difficultys = [0, 1, 2]
    ret = eval_class(gt_annos, dt_annos, current_classes, difficultys, 0, min_overlaps, compute_aos)
    mAP_bbox, mAP_bbox_R40 = get_mAP(ret["precision"]), get_mAP_R40(ret["precision"])
    if PR_detail_dict is not None:
        PR_detail_dict["bbox"] = ret["precision"]
    mAP_aos = mAP_aos_R40 = None
    if compute_aos:
        mAP_aos, mAP_aos_R40 = get_mAP(ret["orientation"]), get_mAP_R40(ret["orientation"])
        if PR_detail_dict is not None:
            PR_detail_dict["aos"] = ret["orientation"]
    ret = eval_class(gt_annos, dt_annos, current_classes, difficultys, 1, min_overlaps)
    mAP_bev, mAP_bev_R40 = get_mAP(ret["precision"]), get_mAP_R40(ret["precision"])
    if PR_detail_dict is not None:
        PR_detail_dict["bev"] = ret["precision"]
    ret = eval_class(gt_annos, dt_annos, current_classes, difficultys, 2, min_overlaps)
    mAP_3d, mAP_3d_R40 = get_mAP(ret["precision"]), get_mAP_R40(ret["precision"])
    if PR_detail_dict is not None:
        PR_detail_dict["3d"] = ret["precision"]
    return mAP_bbox, mAP_bev, mAP_3d, mAP_aos, mAP_bbox_R40, mAP_bev_R40, mAP_3d_R40, mAP_aos_R40


def get_official_eval_result(gt_annos, dt_annos, current_classes, PR_detail_dict=None):
    overlap_0_7 = np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.7], [0.7, 0.5, 0.5, 0.7, 0.5, 0.7], [0.7, 0.5, 0.5, 0.7, 0.5, 0.7]])
    min_overlaps = overlap_0_7[np.newaxis, :, :]
    name_to_class = {v: n for n, v in CLASS_TO_NAME.items()}
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    current_classes = [name_to_class[c] if isinstance(c, str) else c for c in current_classes]
    min_overlaps = min_overlaps[:, :, current_classes]
    compute_aos = False
    for anno in dt_annos:                 # alpha == -10 marks "no orientation estimate" (eval.py:666-671)
        if anno["alpha"].shape[0] != 0:
            if anno["alpha"][0] != -10:
                compute_aos = True
            break
    (mAPbbox, mAPbev, mAP3d, mAPaos, mAPbbox_R40, mAPbev_R40, mAP3d_R40, mAPaos_R40) = do_eval(
        gt_annos, dt_annos, current_classes, min_overlaps, compute_aos, PR_detail_dict=PR_detail_dict)
    lines, ret = [], {}
    diffs = ("easy", "moderate", "hard")

    def triple(tag, a, j, i, fmt):
        lines.append(tag + ", ".join(fmt.format(a[j, d, i]) for d in range(3)))

    for j, curcls in enumerate(current_classes):
        nm = CLASS_TO_NAME[curcls]
        for i in range(min_overlaps.shape[0]):
            head = "{:.2f}, {:.2f}, {:.2f}:".format(*min_overlaps[i, :, j])
            for suffix, (bb, bv, d3, ao) in (("", (mAPbbox, mAPbev, mAP3d, mAPaos)),
                                             ("_R40", (mAPbbox_R40, mAPbev_R40, mAP3d_R40, mAPaos_R40))):
                lines.append("%s AP%s@%s" % (nm, suffix, head))
                triple("bbox AP:", bb, j, i, "{:.4f}")
                triple("bev  AP:", bv, j, i, "{:.4f}")
                triple("3d   AP:", d3, j, i, "{:.4f}")
                if compute_aos:
                    triple("aos  AP:", ao, j, i, "{:.2f}")
                if i == 0:
                    for d, dn in enumerate(diffs):
                        if compute_aos:
                            ret["%s_aos_%s%s" % (nm, dn, suffix)] = ao[j, d, 0]
                        ret["%s_3d_%s%s" % (nm, dn, suffix)] = d3[j, d, 0]
                        ret["%s_bev_%s%s" % (nm, dn, suffix)] = bv[j, d, 0]
                        ret["%s_image_%s%s" % (nm, dn, suffix)] = bb[j, d, 0]
    return "\n".join(lines) + ("\n" if lines else ""), ret
