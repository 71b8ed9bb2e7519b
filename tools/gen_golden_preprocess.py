#!/usr/bin/env python
"""Golden vectors for the test-time input path: the reference's own ``lib.augmentations.Preprocess``
(ConvertToFloat -> Padding -> Normalize, augmentations.py:36-57,128-160,472-501) followed by the BGR->RGB swap and the
HWC->CHW permute of ``lib/dataloader.py:943-950``, run here on seeded uint8 frames.

cv2 is not installed in the build container: the two cv2 calls on this path are given functional numpy stand-ins with the
documented OpenCV semantics (``copyMakeBorder(..., BORDER_CONSTANT, value=[0,0,0])`` = zero padding at the bottom / right,
``cvtColor(COLOR_BGR2RGB)`` = channel reversal); everything else is the reference's code.
Writes tests/golden/preprocess.npz (inputs + expected outputs; data only)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden  # noqa: E402


def main():
    gen_golden._install_stubs()
    cv2 = sys.modules["cv2"]
    cv2.BORDER_CONSTANT = 0
    cv2.COLOR_BGR2RGB = 4
    cv2.copyMakeBorder = lambda im, top, bottom, left, right, kind, value=None: np.pad(
        im, ((top, bottom), (left, right), (0, 0)), mode="constant", constant_values=0)
    cv2.cvtColor = lambda im, code: np.ascontiguousarray(im[:, :, ::-1])
    import torch
    from lib.augmentations import Preprocess
    mean, stds = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]          # scripts/config/kitti_3d_anab.py:42-43
    out = {}
    rng = np.random.RandomState(7)
    for name, (h, w), size in (("a", (60, 150), (64, 160)), ("b", (64, 160), (64, 160)), ("c", (37, 53), (64, 96))):
        im = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        im[0, 0] = (0, 128, 255)
        pre = Preprocess(size, mean, stds)
        x, _ = pre(im.copy(), None)
        x = cv2.cvtColor(x, cv2.COLOR_BGR2RGB)                         # dataloader.py:943
        x = torch.from_numpy(x).permute(2, 0, 1).contiguous().numpy()  # dataloader.py:950
        assert x.dtype == np.float32 and x.shape == (3,) + tuple(size)
        out["in_" + name], out["out_" + name], out["size_" + name] = im, x, np.asarray(size)
    out["mean"], out["stds"] = np.asarray(mean, np.float32), np.asarray(stds, np.float32)
    path = os.path.join(gen_golden.OUT, "preprocess.npz")
    np.savez_compressed(path, **out)
    print(path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
