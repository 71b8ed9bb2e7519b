"""Drop-in for the label / result readers of the reference's lib/eval/kitti_common.py."""
from m3dssd_amd.eval.kitti_common import (filter_annos_low_score, get_image_index_str, get_label_anno,  # noqa: F401
                                          get_label_annos)
