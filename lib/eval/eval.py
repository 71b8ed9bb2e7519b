"""Drop-in for the reference's lib/eval/eval.py (get_official_eval_result and the pieces callers import)."""
from m3dssd_amd.eval.eval import (bev_box_overlap, calculate_iou_partly, clean_data, d3_box_overlap, do_eval, eval_class,  # noqa: F401
                                  get_mAP, get_mAP_R40, get_official_eval_result, get_split_parts, get_thresholds,
                                  image_box_overlap)
