"""Drop-in for the reference's lib/eval/rotate_iou.py: ``rotate_iou_gpu_eval`` on the HIP kernel."""
from m3dssd_amd.eval.eval import rotate_iou_eval


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    """lib/eval/rotate_iou.py:264-326: float [N,5] x [K,5] -> [N,K] in the dtype of `boxes`."""
    import numpy as np
    return rotate_iou_eval(boxes, query_boxes, criterion, device="cuda:%d" % device_id).astype(np.asarray(boxes).dtype)
