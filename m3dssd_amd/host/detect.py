"""Post-forward detection: decode -> top-N-pre -> NMS -> row selection, on the device.

``im_detect_3d`` keeps the reference's signature and row format (lib/rpn_util.py:1416-1563, the
``synced=False`` branch): returns float32 ndarray [K, 14] =
x1,y1,x2,y2,score,cls,x3d,y3d,z3d,w3d,h3d,l3d,ry3d,anchor in descending-score order.
``detect_batch`` is the batched form the reference lacks (it indexes batch 0, :1484-1503): it
returns fixed-size blocks [B, nms_topN_post, 14] + counts, the wire format of the multi-GPU gather.

The score sort uses the 64-bit keys written by ``m3d_bundle_outputs`` (score bits, inverted row id) so
the order is total: descending score, ascending row among equals."""
import ctypes

import numpy as np
import torch

from .. import _hip


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def detect_from_outputs(eng, plan, prob, bbox_2d, bbox_3d, rois, conf):
    """decode -> top-N-pre -> NMS on the engine's output buffers (current stream).
    -> (aboxes [B, n_pre, 14] score-sorted, keep [B, n_pre] int32 positions, num_keep [B] int32)."""
    L = _hip.lib()
    dev = prob.device
    B, R = prob.shape[0], prob.shape[1]
    keys = plan.named["score_key"]
    n_pre = min(int(conf.nms_topN_pre), R)
    top = torch.topk(keys, n_pre, dim=1, largest=True, sorted=True)[0]
    rows = (0xFFFFFFFF - (top & 0xFFFFFFFF)).contiguous()              # int64 row ids, score-descending
    aboxes = torch.empty(B, n_pre, 14, device=dev, dtype=torch.float32)
    P = eng.P
    with torch.cuda.device(dev):
        _hip.check(L.m3d_decode_rows(rows.data_ptr(), prob.data_ptr(), bbox_2d.data_ptr(), bbox_3d.data_ptr(),
                                     rois.data_ptr(), P["anchors"].data_ptr(), P["means"].data_ptr(),
                                     P["stds"].data_ptr(), aboxes.data_ptr(), B, R, n_pre, _stream()))
        keep = torch.empty(B, n_pre, device=dev, dtype=torch.int32)
        num = torch.zeros(B, device=dev, dtype=torch.int32)
        ws = torch.empty(L.m3d_nms_workspace_bytes(B, n_pre), device=dev, dtype=torch.uint8)
        _hip.check(L.m3d_nms_sorted_dev(aboxes.data_ptr(), B, n_pre, 14, float(conf.nms_thres), ws.data_ptr(),
                                        keep.data_ptr(), num.data_ptr(), _stream()))
    return aboxes, keep, num


def detect_device(net, im, conf, top_post=None):
    """-> (aboxes [B, n_pre, 14] score-sorted, keep [B, n_pre] int32 positions, num_keep [B] int32), device tensors."""
    if im.dim() == 3:
        im = im[None]
    dev = next(net.parameters()).device
    im = im.to(dev, torch.float32)
    with torch.no_grad():
        net.eval()
        cls, prob, bbox_2d, bbox_3d, feat_size, rois = net(im)
        eng = net.engine()
        plan = eng.plan_for(prob.shape[0], im.shape[2], im.shape[3])
        return detect_from_outputs(eng, plan, prob, bbox_2d, bbox_3d, rois, conf)


def im_detect_3d(im, net, rpn_conf, obj=None, gpu=0, synced=False):
    if synced:
        raise NotImplementedError("the synced=True branch of im_detect_3d is not used by test_kitti_3d")
    aboxes, keep, num = detect_device(net, im, rpn_conf)
    k = int(num[0].item())
    out = aboxes[0][keep[0, :k].long()].cpu().numpy()
    scale = getattr(obj, "scale_factor", 1.0) if obj is not None else 1.0
    if scale != 1.0:
        out[:, 0:4] /= scale
        out[:, 6:8] /= scale
    if rpn_conf.clip_boxes and obj is not None:
        out[:, 0] = np.clip(out[:, 0], 0, obj.imW - 1)
        out[:, 1] = np.clip(out[:, 1], 0, obj.imH - 1)
        out[:, 2] = np.clip(out[:, 2], 0, obj.imW - 1)
        out[:, 3] = np.clip(out[:, 3], 0, obj.imH - 1)
    return out


def select_post(aboxes, keep, num, conf):
    """Kept rows -> fixed-size blocks (dets [B, nms_topN_post, 14] zero-padded, counts [B] int32)."""
    B = aboxes.shape[0]
    post = int(conf.nms_topN_post)
    idx = keep[:, :post].long().clamp_(min=0, max=aboxes.shape[1] - 1)
    counts = torch.clamp(num, max=post)
    dets = torch.gather(aboxes, 1, idx[:, :, None].expand(B, idx.shape[1], 14))
    valid = torch.arange(idx.shape[1], device=dets.device)[None, :] < counts[:, None]
    dets = dets * valid[:, :, None].to(dets.dtype)
    if dets.shape[1] < post:
        dets = torch.cat([dets, dets.new_zeros(B, post - dets.shape[1], 14)], 1)
    return dets.contiguous(), counts.to(torch.int32)


def detect_batch(net, im, conf):
    """-> (dets [B, nms_topN_post, 14] zero-padded, counts [B] int32) device tensors."""
    return select_post(*detect_device(net, im, conf), conf)
