"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's test-time input path.

    lib/augmentations.py:36-41   ConvertToFloat   image.astype(float32)
    lib/augmentations.py:138-160 Padding          zero border at the bottom / right up to `size` (applied to the RAW image)
    lib/augmentations.py:44-57   Normalize        image /= 255.0; image -= mean; image /= stds   (float32, in this order,
                                                  mean / stds indexed by the channel POSITION of the BGR image)
    lib/dataloader.py:943-950    BGR -> RGB, HWC -> CHW

so network channel c = ((u8[2 - c] / 255) - mean[2 - c]) / stds[2 - c] inside the original image and
(0 - mean[2 - c]) / stds[2 - c] in the padded border.  Pinned by tests/golden/preprocess.npz (tools/gen_golden_preprocess.py).
"""
import numpy as np


def preprocess(img_u8_bgr, size, mean, stds):
    """img [h, w, 3] uint8 (BGR, as cv2.imread returns it) -> [3, size[0], size[1]] float32 (RGB planes)."""
    img = np.asarray(img_u8_bgr)
    h, w, c = img.shape
    if c != 3 or h > size[0] or w > size[1]:
        raise ValueError("preprocess: need an HxWx3 image no larger than the target size")
    x = np.zeros((size[0], size[1], 3), dtype=np.float32)
    x[:h, :w] = img.astype(np.float32)
    x /= np.float32(255.0)
    x -= np.asarray(mean, dtype=np.float32)
    x /= np.asarray(stds, dtype=np.float32)
    return np.ascontiguousarray(x[:, :, ::-1].transpose(2, 0, 1))
