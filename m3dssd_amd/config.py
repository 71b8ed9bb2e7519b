"""Configuration object for the M3DSSD inference path.

The reference passes an ``EasyDict`` (scripts/config/kitti_3d_anab_fullalign.py:4-152,
re-read from a pickle at test time, scripts/test_rpn_3d.py:27).  ``Conf`` gives the
same access patterns (attribute, ``in``, ``[]``) without the easydict dependency;
``Config()`` fills the fields RPN/DLASeg read (M3d_inference_align.py:41-63,138-168,
pose_dla_dcn.py:529) with the shipped values, except ``back_bone`` which this path
fixes to ``dla34`` (BASELINE.json north_star).
"""
import numpy as np


class Conf(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def copy(self):
        return Conf(dict.copy(self))


def Config():
    conf = Conf()
    conf.model = "M3d_inference_align"
    conf.ida_dcnv2 = True
    conf.attention = "ANAB"
    conf.center_align = True
    conf.shape_align = True
    conf.image_means = [0.485, 0.456, 0.406]
    conf.image_stds = [0.229, 0.224, 0.225]
    conf.feat_stride = 8
    conf.back_bone = "dla34"
    conf.pre_train = False
    conf.test_scale = [384, 1280]
    conf.crop_size = [384, 1280]
    conf.percent_anc_h = [0.0625, 0.75]
    conf.min_gt_h = conf.test_scale[0] * conf.percent_anc_h[0]
    conf.max_gt_h = conf.test_scale[0] * conf.percent_anc_h[1]
    conf.lbls = ["Car", "Pedestrian", "Cyclist"]
    conf.ilbls = ["Van", "ignore"]
    conf.batch_size = 4
    conf.nms_topN_pre = 3000
    conf.nms_topN_post = 40
    conf.nms_thres = 0.4
    conf.clip_boxes = False
    conf.cluster_anchors = 0
    conf.anchors = None
    conf.bbox_means = None
    conf.bbox_stds = None
    base = (conf.max_gt_h / conf.min_gt_h) ** (1 / (12 - 1))
    conf.anchor_scales = np.array([conf.min_gt_h * (base ** i) for i in range(0, 12)])
    conf.anchor_ratios = np.array([0.5, 1.0, 1.5])
    conf.device = "cuda:0"
    return conf
