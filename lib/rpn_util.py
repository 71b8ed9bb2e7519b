"""Drop-in for the inference part of the reference's lib/rpn_util.py."""
from m3dssd_amd.rpn_util import (anchor_center, generate_anchors_2d, calc_output_size, locate_anchors,  # noqa: F401
                                 flatten_tensor)
from m3dssd_amd.host.detect import im_detect_3d, detect_batch  # noqa: F401
from m3dssd_amd.host.nms import gpu_nms  # noqa: F401
from m3dssd_amd.host.kitti_test import test_kitti_3d  # noqa: F401,E402
