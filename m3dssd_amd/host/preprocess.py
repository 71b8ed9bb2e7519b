"""Test-time input path on the device.

``preprocess(frames, size, mean, stds)`` = the reference's ``lib.augmentations.Preprocess`` (ConvertToFloat -> Padding ->
Normalize, augmentations.py:36-57,138-160,472-501) followed by the BGR->RGB swap and HWC->CHW permute of
``lib/dataloader.py:943-950``, for a batch of uint8 BGR frames already in device memory; bit-identical to the numpy
arithmetic (tests/golden/preprocess.npz).  ``RPN.forward`` also accepts the uint8 frames directly, in which case the same
arithmetic runs inside the stem kernel's loads and no float image is written at all.
"""
import ctypes

import torch

from .. import _hip


def preprocess(frames, size, mean, stds):
    """frames: uint8 [B, h, w, 3] (or [h, w, 3]) BGR ROCm tensor -> float32 [B, 3, size[0], size[1]] RGB planes."""
    if not isinstance(frames, torch.Tensor) or not frames.is_cuda:
        raise NotImplementedError("preprocess: ROCm device tensor expected")          # same stance as dcn_v2_func.py:23-24
    if frames.dtype != torch.uint8 or frames.shape[-1] != 3 or frames.dim() not in (3, 4):
        raise RuntimeError("preprocess: uint8 [B, h, w, 3] frames expected")
    if frames.dim() == 3:
        frames = frames[None]
    frames = frames.contiguous()
    B, h, w, _ = frames.shape
    H, W = int(size[0]), int(size[1])
    out = torch.empty(B, 3, H, W, device=frames.device, dtype=torch.float32)
    m3 = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s3 = (ctypes.c_float * 3)(*[float(v) for v in stds])
    with torch.cuda.device(frames.device):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _hip.check(_hip.lib().m3d_preprocess_u8(frames.data_ptr(), B, h, w, m3, s3, out.data_ptr(), H, W, st))
    return out
