"""KITTI AP evaluation on the MI355X path: label I/O (kitti_common), the official AP / AP_R40 tables
(eval) with the rotated-IoU matrix computed by a HIP kernel and the greedy matching in native host code."""
from .eval import get_official_eval_result  # noqa: F401
from .kitti_common import get_label_anno, get_label_annos  # noqa: F401
