"""KITTI label / result file reader with the reference's annotation dict layout (lib/eval/kitti_common.py:293-345):
name, truncated, occluded, alpha, bbox [N,4], dimensions [N,3] (hwl in the file -> lhw), location [N,3], rotation_y, score."""
import pathlib
import re

import numpy as np


def get_image_index_str(img_idx):
    return "{:06d}".format(img_idx)


def get_label_anno(label_path):
    with open(label_path, "r") as f:
        rows = [line.strip().split(" ") for line in f.readlines()]
    n = len(rows)
    num = np.array([[float(v) for v in r[1:15]] for r in rows], dtype=np.float64).reshape(n, 14)
    anno = {
        "name": np.array([r[0] for r in rows]),
        "truncated": num[:, 0].copy(),
        "occluded": np.array([int(r[2]) for r in rows]),
        "alpha": num[:, 2].copy(),
        "bbox": num[:, 3:7].copy(),
        "dimensions": num[:, 7:10][:, [2, 0, 1]].copy(),      # file order h, w, l -> l, h, w (camera)
        "location": num[:, 10:13].copy(),
        "rotation_y": num[:, 13].copy(),
    }
    if n != 0 and len(rows[0]) == 16:
        anno["score"] = np.array([float(r[15]) for r in rows])
    else:
        anno["score"] = np.zeros([n])
    return anno


def get_label_annos(label_folder, image_ids=None):
    folder = pathlib.Path(label_folder)
    if image_ids is None:
        pat = re.compile(r"^\d{6}.txt$")
        image_ids = sorted(int(p.stem) for p in folder.glob("*.txt") if pat.match(p.name))
    if not isinstance(image_ids, list):
        image_ids = list(range(image_ids))
    return [get_label_anno(folder / (get_image_index_str(i) + ".txt")) for i in image_ids]


def filter_annos_low_score(image_annos, thresh):
    out = []
    for anno in image_annos:
        keep = [i for i, s in enumerate(anno["score"]) if s >= thresh]
        out.append({k: v[keep] for k, v in anno.items()})
    return out
