"""Tensor-level wrappers over the C ABI for NCHW torch tensors (the op-level drop-in boundary).

These are what the standalone modules (DCNv2, DCN, ...) call.  The whole-network path
(``RPN.forward``) does not go through here -- it runs the NHWC engine (m3dssd_amd/engine.py).
"""
import ctypes

import torch

from .. import _hip


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_cuda(*ts):
    for t in ts:
        if not t.is_cuda:
            # the reference raises NotImplementedError for non-CUDA input (model/DCNv2/dcn_v2_func.py:23-24)
            raise NotImplementedError("M3DSSD HIP ops need ROCm device tensors; there is no CPU fallback")


def dcn_v2_forward(inp, offset, mask, weight, bias, stride, padding, dilation=1, deformable_groups=1):
    """DCNv2Function.forward (model/DCNv2/dcn_v2_func.py:22-38) on the HIP library."""
    _require_cuda(inp, offset, mask, weight, bias)
    if not inp.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")          # dcn_v2_cuda.c:21
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")         # dcn_v2_cuda.c:22
    if inp.dtype != torch.float32:
        raise RuntimeError("dcn_v2_forward: float32 only")
    L = _hip.lib()
    n, c, h, w = inp.shape
    co, ck, kh, kw = weight.shape
    if ck != c:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)." % (c, ck))   # dcn_v2_cuda.c:37-39
    if deformable_groups < 1 or c % deformable_groups:
        raise RuntimeError("dcn_v2_forward: deformable_groups (%d) must divide the input channels (%d)" % (deformable_groups, c))
    ho = (h + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    wo = (w + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    if tuple(offset.shape) != (n, deformable_groups * 2 * kh * kw, ho, wo) or \
            tuple(mask.shape) != (n, deformable_groups * kh * kw, ho, wo):
        raise RuntimeError("dcn_v2_forward: offset/mask shape does not match the output size")
    offset, mask, bias = offset.contiguous().float(), mask.contiguous().float(), bias.contiguous().float()
    out = torch.empty(n, co, ho, wo, device=inp.device, dtype=torch.float32)
    nbytes = L.m3d_dcn_v2_workspace_bytes_grouped(n, c, h, w, co, kh, kw, stride, padding, dilation, deformable_groups)
    ws = torch.empty(nbytes + 256, device=inp.device, dtype=torch.uint8)
    base = (ws.data_ptr() + 255) // 256 * 256
    with torch.cuda.device(inp.device):
        _hip.check(L.m3d_dcn_v2_forward(inp.data_ptr(), weight.data_ptr(), bias.data_ptr(), offset.data_ptr(),
                                        mask.data_ptr(), out.data_ptr(), n, c, h, w, co, kh, kw, stride, stride,
                                        padding, padding, dilation, dilation, deformable_groups, base, nbytes, _stream()))
    return out


def nms_sorted(boxes_sorted, thresh):
    """Device NMS on score-sorted boxes [B, n, >=4] (or [n, >=4]) -> (keep [B, n] int32, num [B] int32)."""
    _require_cuda(boxes_sorted)
    L = _hip.lib()
    b3 = boxes_sorted if boxes_sorted.dim() == 3 else boxes_sorted[None]
    b3 = b3.contiguous().float()
    B, n, s = b3.shape
    keep = torch.empty(B, max(n, 1), device=b3.device, dtype=torch.int32)
    num = torch.zeros(B, device=b3.device, dtype=torch.int32)
    ws = torch.empty(max(1, L.m3d_nms_workspace_bytes(B, n)), device=b3.device, dtype=torch.uint8)
    with torch.cuda.device(b3.device):
        _hip.check(L.m3d_nms_sorted_dev(b3.data_ptr(), B, n, s, float(thresh), ws.data_ptr(), keep.data_ptr(),
                                        num.data_ptr(), _stream()))
    return keep, num
