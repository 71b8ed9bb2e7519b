"""ORACLE (test infrastructure): DCNv2 forward on the CPU.

Restates model/DCNv2/dcn_v2_func.py:22-38 (+ _infer_shape :64-73) and
model/DCNv2/src/dcn_v2_cuda.c:10-102: per image  out = bias (x) ones  (:72-78),
columns = modulated_deformable_im2col(...) (:80-87, C restatement in
oracle/dcn_im2col.c), out += W[Co, C*kh*kw] @ columns (:90-96, BLAS like the
reference's cuBLAS Sgemm).  ``dcn_v2_forward_numpy`` is an independent slow
pure-numpy restatement used to cross-check the C one on tiny cases.
"""
import ctypes
import math

import numpy as np
import torch

from . import _clib


def out_size(h, w, kh, kw, stride, pad, dil):
    # dcn_v2_cuda.c:40-41 / dcn_v2_func.py:69-72
    ho = (h + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    wo = (w + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    return ho, wo


def dcn_v2_forward(inp, offset, mask, weight, bias, stride=1, pad=0, dil=1, deformable_groups=1):
    """inp [N,C,H,W], offset [N,dg*2*kh*kw,Ho,Wo], mask [N,dg*kh*kw,Ho,Wo],
    weight [Co,C,kh,kw], bias [Co] -> [N,Co,Ho,Wo]; float32 CPU tensors."""
    if inp.is_cuda:
        raise RuntimeError("oracle runs on the CPU only")
    inp = inp.detach().contiguous().float()
    offset = offset.detach().contiguous().float()
    mask = mask.detach().contiguous().float()
    weight = weight.detach().contiguous().float()
    bias = bias.detach().contiguous().float()
    n, c, h, w = inp.shape
    co, ck, kh, kw = weight.shape
    if ck != c:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)." % (c, ck))
    ho, wo = out_size(h, w, kh, kw, stride, pad, dil)
    assert offset.shape == (n, deformable_groups * 2 * kh * kw, ho, wo), offset.shape
    assert mask.shape == (n, deformable_groups * kh * kw, ho, wo), mask.shape
    L = _clib.lib()
    fp = ctypes.POINTER(ctypes.c_float)
    out = torch.empty(n, co, ho, wo, dtype=torch.float32)
    col = torch.empty(c * kh * kw, ho * wo, dtype=torch.float32)
    wmat = weight.view(co, c * kh * kw)
    for b in range(n):
        L.oracle_dcn_im2col(
            ctypes.cast(inp[b].data_ptr(), fp), ctypes.cast(offset[b].data_ptr(), fp),
            ctypes.cast(mask[b].data_ptr(), fp), c, h, w, ho, wo, kh, kw, pad, pad,
            stride, stride, dil, dil, deformable_groups, ctypes.cast(col.data_ptr(), fp))
        out[b] = torch.addmm(bias.view(co, 1).expand(co, ho * wo), wmat, col).view(co, ho, wo)
    return out


def dcn_v2_forward_numpy(inp, offset, mask, weight, bias, stride=1, pad=0, dil=1):
    """Slow independent restatement (deformable_groups=1), float64 accumulation.
    dcn_v2_im2col_cuda.cu:18-47,129-178 written as per-output-pixel loops."""
    inp, offset, mask = (np.asarray(a, dtype=np.float32) for a in (inp, offset, mask))
    weight, bias = np.asarray(weight, dtype=np.float32), np.asarray(bias, dtype=np.float32)
    n, c, h, w = inp.shape
    co, _, kh, kw = weight.shape
    ho, wo = out_size(h, w, kh, kw, stride, pad, dil)
    out = np.zeros((n, co, ho, wo), dtype=np.float64)
    for b in range(n):
        for y in range(ho):
            for x in range(wo):
                acc = bias.astype(np.float64).copy()
                for i in range(kh):
                    for j in range(kw):
                        k = i * kw + j
                        dh = np.float32(offset[b, 2 * k, y, x])
                        dw = np.float32(offset[b, 2 * k + 1, y, x])
                        m = np.float32(mask[b, k, y, x])
                        h_im = np.float32(y * stride - pad + i * dil) + dh
                        w_im = np.float32(x * stride - pad + j * dil) + dw
                        if not (h_im > -1 and w_im > -1 and h_im < h and w_im < w):
                            continue
                        hl, wl = int(math.floor(h_im)), int(math.floor(w_im))
                        hh_, wh_ = hl + 1, wl + 1
                        lh, lw = np.float32(h_im - np.float32(hl)), np.float32(w_im - np.float32(wl))
                        hh, hw = np.float32(1) - lh, np.float32(1) - lw
                        z = np.zeros(c, dtype=np.float32)
                        v1 = inp[b, :, hl, wl] if (hl >= 0 and wl >= 0) else z
                        v2 = inp[b, :, hl, wh_] if (hl >= 0 and wh_ <= w - 1) else z
                        v3 = inp[b, :, hh_, wl] if (hh_ <= h - 1 and wl >= 0) else z
                        v4 = inp[b, :, hh_, wh_] if (hh_ <= h - 1 and wh_ <= w - 1) else z
                        val = (hh * hw) * v1 + (hh * lw) * v2 + (lh * hw) * v3 + (lh * lw) * v4
                        val = (val * m).astype(np.float64)
                        acc += weight[:, :, i, j].astype(np.float64) @ val
                out[b, :, y, x] = acc
    return out.astype(np.float32)
