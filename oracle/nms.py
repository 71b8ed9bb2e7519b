"""ORACLE (test infrastructure): greedy NMS on the CPU.

``gpu_nms`` mirrors the reference wrapper lib/nms/gpu_nms.pyx:16-31 (sort by
descending score, call _nms, map kept positions back through ``order``) on top of
the C restatement of lib/nms/nms_kernel.cu (oracle/nms_ref.c).  ``nms_numpy``
restates lib/nms/py_cpu_nms.py:10-38 (same "+1" areas, keep iff ovr <= thresh)
and is what the golden vectors from the reference's py_cpu_nms are compared with.

Tie order: the reference's ``scores.argsort()[::-1]`` (gpu_nms.pyx:26) is an
unstable sort reversed, i.e. undefined among equal scores.  This build DEFINES
the order as: descending score, ascending original index among equals (stable);
golden fixtures use distinct scores so both definitions agree.
"""
import ctypes

import numpy as np

from . import _clib


def order_desc_stable(scores):
    scores = np.asarray(scores)
    return np.argsort(-scores.astype(np.float64), kind="stable")


def nms_sorted(sorted_dets, thresh):
    """_nms restated (nms_kernel.cu:91-144): input already sorted; returns kept positions."""
    d = np.ascontiguousarray(sorted_dets, dtype=np.float32)
    n, dim = d.shape
    keep = np.zeros(max(n, 1), dtype=np.int32)
    num = ctypes.c_int(0)
    if n:
        _clib.lib().oracle_nms_sorted(
            keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(num),
            d.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n, dim, ctypes.c_float(thresh))
    return keep[:num.value]


def gpu_nms(dets, thresh, device_id=0):
    """gpu_nms.pyx:16-31 restated; returns list of indices into ``dets``."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    order = order_desc_stable(dets[:, 4])
    keep = nms_sorted(dets[order, :], thresh)
    return list(order[keep])


def nms_numpy(dets, thresh):
    """py_cpu_nms.py:10-38 restated (vectorised per kept box), stable tie order."""
    dets = np.asarray(dets, dtype=np.float32)
    x1, y1, x2, y2 = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = order_desc_stable(dets[:, 4])
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        rest = order[1:]
        w = np.maximum(np.float32(0.0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1)
        h = np.maximum(np.float32(0.0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        order = rest[ovr <= thresh]
    return keep
