"""``gpu_nms`` with the reference's signature (lib/nms/gpu_nms.pyx:16-31): float32 [N, >=5] host
array, threshold, device id -> python list of kept indices into ``dets`` in descending-score order.

Score order: descending score, ascending index among equals (a stable definition of the reference's
``scores.argsort()[::-1]``, which is unstable; identical for distinct scores).  The boxes go to the
device once, sorted; masks + greedy reduction run on the device; only the keep list comes back."""
import ctypes

import numpy as np

from .. import _hip


def gpu_nms(dets, thresh, device_id=0):
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    if dets.ndim != 2 or dets.shape[1] < 5:
        raise ValueError("dets must be [N, >=5] (x1, y1, x2, y2, score)")
    n, dim = dets.shape
    if n == 0:
        return []
    order = np.argsort(-dets[:, 4].astype(np.float64), kind="stable")
    sorted_dets = np.ascontiguousarray(dets[order, :])
    keep = np.zeros(n, dtype=np.int32)
    num = ctypes.c_int(0)
    _hip.lib()._nms(keep.ctypes.data_as(ctypes.c_void_p), ctypes.cast(ctypes.byref(num), ctypes.c_void_p),
                    sorted_dets.ctypes.data_as(ctypes.c_void_p), n, dim, float(thresh), int(device_id))
    return list(order[keep[:num.value]])
