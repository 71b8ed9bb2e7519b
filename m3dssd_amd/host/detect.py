"""Post-forward detection: decode -> top-N-pre -> NMS -> row selection, on the device.

``im_detect_3d`` keeps the reference's signature and row format (lib/rpn_util.py:1416-1563, the
``synced=False`` branch): returns float32 ndarray [K, 14] =
x1,y1,x2,y2,score,cls,x3d,y3d,z3d,w3d,h3d,l3d,ry3d,anchor in descending-score order.
``detect_batch`` is the batched form the reference lacks (it indexes batch 0, :1484-1503): it
returns fixed-size blocks [B, nms_topN_post, 14] + counts, the wire format of the multi-GPU gather.

The descending score sort + top-N-pre cut + decode is ONE launch (``m3d_topk_decode``: radix select over the total
order "descending score, ascending row among equals" on the score bits written by ``m3d_bundle_outputs``), NMS is two
(``m3d_nms_sorted_dev``), the post-NMS row selection one (``m3d_select_post``): four launches, no ATen kernels."""
import ctypes

import numpy as np
import torch

from .. import _hip

NMS_MAX_PRE = 16384       # m3d_nms_sorted_dev / m3d_topk_decode: "removed" words and the k sort keys live in LDS


def unwrap(net):
    """The RPN module behind the wrappers the reference's scripts put around it: `nn.DataParallel(net)`
    (scripts/test_rpn_3d.py:50-51, lib/core.py:73-74) or DistributedDataParallel hand `.module`.  The device detection stage needs
    the module itself (its packed engine and plan buffers); with one process per GPU (m3dssd_amd.dist) that is also the whole of
    what the wrapper would have done."""
    seen = 0
    while not hasattr(net, "engine") and hasattr(net, "module") and seen < 4:
        net, seen = net.module, seen + 1
    if not hasattr(net, "engine"):
        raise TypeError("expected the RPN module of model.M3d_inference_align.build() (or a DataParallel / DDP wrapper of it), got %s"
                        % type(net).__name__)
    return net


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def check_conf(conf):
    """Limits of the device detection stage, checked where the module is built (clear message instead of a C-ABI error)."""
    if int(conf.nms_topN_pre) > NMS_MAX_PRE:
        raise ValueError("nms_topN_pre = %d: the device NMS / top-k handle at most %d pre-NMS boxes per image "
                         "(the reference default is 3000)" % (int(conf.nms_topN_pre), NMS_MAX_PRE))
    if len(conf.lbls) + 1 != 4:
        raise ValueError("the output bundling / decode kernels are written for 4 classes (background + 3), got %d"
                         % (len(conf.lbls) + 1))


def _scale_tensor(scale, B, dev):
    """scale: None, a float, or [B] floats / tensor -> device float32 [B] (or None when every factor is 1)."""
    if scale is None:
        return None
    if torch.is_tensor(scale):
        return scale.to(dev, torch.float32).reshape(B).contiguous()
    arr = np.broadcast_to(np.asarray(scale, dtype=np.float32).reshape(-1), (B,)).copy()
    if (arr == 1.0).all():
        return None
    return torch.from_numpy(arr).to(dev)


def detect_from_outputs(eng, plan, prob, bbox_2d, bbox_3d, rois, conf, scale=None):
    """top-N-pre select + decode -> NMS on the engine's output buffers (current stream of their device).
    -> (aboxes [B, n_pre, 14] score-sorted, keep [B, n_pre] int32 positions, num_keep [B] int32).
    scale: per-image test-time scale factors (device float32 [B]) -- the boxes are divided by them BEFORE the NMS like the
    reference's im_detect_3d does (lib/rpn_util.py:1504-1506)."""
    L = _hip.lib()
    dev = prob.device
    B, R = prob.shape[0], prob.shape[1]
    bits = plan.named["score_bits"]
    n_pre = min(int(conf.nms_topN_pre), R)
    P = eng.P
    with torch.cuda.device(dev):
        st = _stream(dev)
        aboxes = torch.empty(B, n_pre, 14, device=dev, dtype=torch.float32)
        keep = torch.empty(B, n_pre, device=dev, dtype=torch.int32)
        num = torch.empty(B, device=dev, dtype=torch.int32)
        tk_bytes = L.m3d_topk_decode_workspace_bytes(B, R)
        ws = torch.empty(max(tk_bytes, L.m3d_nms_workspace_bytes(B, n_pre)), device=dev, dtype=torch.uint8)
        _hip.check(L.m3d_topk_decode_scaled(bits.data_ptr(), prob.data_ptr(), bbox_2d.data_ptr(), bbox_3d.data_ptr(),
                                            rois.data_ptr(), P["anchors"].data_ptr(), P["means"].data_ptr(),
                                            P["stds"].data_ptr(), None if scale is None else scale.data_ptr(), aboxes.data_ptr(),
                                            None, ws.data_ptr(), tk_bytes, B, R, n_pre, st))
        _hip.check(L.m3d_nms_sorted_dev(aboxes.data_ptr(), B, n_pre, 14, float(conf.nms_thres), ws.data_ptr(),
                                        keep.data_ptr(), num.data_ptr(), st))
    return aboxes, keep, num


def detect_from_planar(eng, plan, rois, conf, scale=None):
    """detect_from_outputs without the bundled tensors: the sort keys come from ``m3d_score_keys_planar`` (run behind the forward
    in place of ``m3d_bundle_outputs``) and the top-N-pre rows are decoded straight from the planar staging the heads write
    (``m3d_topk_decode_planar``).  Same rows, same bits (tests/test_gpu_detect.py); 5.5 MB per image instead of 38."""
    L = _hip.lib()
    cls_pl, box_pl, bits = plan.named["cls_planar"], plan.named["box_planar"], plan.named["score_bits"]
    dev = bits.device
    B, R = bits.shape[0], bits.shape[1]
    A = eng.A
    HW = R // A
    n_pre = min(int(conf.nms_topN_pre), R)
    P = eng.P
    with torch.cuda.device(dev):
        st = _stream(dev)
        aboxes = torch.empty(B, n_pre, 14, device=dev, dtype=torch.float32)
        keep = torch.empty(B, n_pre, device=dev, dtype=torch.int32)
        num = torch.empty(B, device=dev, dtype=torch.int32)
        tk_bytes = L.m3d_topk_decode_workspace_bytes(B, R)
        ws = torch.empty(max(tk_bytes, L.m3d_nms_workspace_bytes(B, n_pre)), device=dev, dtype=torch.uint8)
        _hip.check(L.m3d_topk_decode_planar(bits.data_ptr(), cls_pl.data_ptr(), box_pl.data_ptr(), rois.data_ptr(),
                                            P["anchors"].data_ptr(), P["means"].data_ptr(), P["stds"].data_ptr(),
                                            None if scale is None else scale.data_ptr(), aboxes.data_ptr(), None, ws.data_ptr(),
                                            tk_bytes, B, A, HW, n_pre, st))
        _hip.check(L.m3d_nms_sorted_dev(aboxes.data_ptr(), B, n_pre, 14, float(conf.nms_thres), ws.data_ptr(),
                                        keep.data_ptr(), num.data_ptr(), st))
    return aboxes, keep, num


def score_keys_planar(eng, plan):
    """The launch that replaces bundle_outputs when only the detection stage follows: sort keys from the planar class logits."""
    bits = plan.named["score_bits"]
    B, R = bits.shape[0], bits.shape[1]
    dev = bits.device
    with torch.cuda.device(dev):
        _hip.check(_hip.lib().m3d_score_keys_planar(plan.named["cls_planar"].data_ptr(), bits.data_ptr(), B, eng.A, R // eng.A,
                                                    _stream(dev)))


def detect_device(net, im, conf, top_post=None, scale=None):
    """-> (aboxes [B, n_pre, 14] score-sorted, keep [B, n_pre] int32 positions, num_keep [B] int32), device tensors.
    scale: test-time scale factor(s) of the frames (float, [B] floats or tensor): applied before the NMS."""
    net = unwrap(net)
    if im.dim() == 3:
        im = im[None]
    dev = next(net.parameters()).device
    u8 = im.dtype == torch.uint8 and im.shape[-1] == 3       # raw BGR frames [B, h, w, 3]: Preprocess runs inside the stem kernel
    im = im.to(dev) if u8 else im.to(dev, torch.float32)
    with torch.no_grad():
        net.eval()
        # plan-owned views, consumed right here (net(im) itself returns fresh tensors: host/rpn.py)
        cls, prob, bbox_2d, bbox_3d, feat_size, rois = net._forward_views(im)
        eng = net.engine()
        H, W = (int(v) for v in conf.crop_size) if u8 else (im.shape[2], im.shape[3])
        plan = eng.plan_for(prob.shape[0], H, W)
        return detect_from_outputs(eng, plan, prob, bbox_2d, bbox_3d, rois, conf, _scale_tensor(scale, prob.shape[0], dev))


def im_detect_3d(im, net, rpn_conf, obj=None, gpu=0, synced=False):
    if synced:
        raise NotImplementedError("the synced=True branch of im_detect_3d is not used by test_kitti_3d")
    scale = getattr(obj, "scale_factor", 1.0) if obj is not None else 1.0
    aboxes, keep, num = detect_device(net, im, rpn_conf, scale=scale)       # scaled before the NMS (lib/rpn_util.py:1504-1506)
    k = int(num[0].item())
    out = aboxes[0][keep[0, :k].long()].cpu().numpy()
    if rpn_conf.clip_boxes and obj is not None:
        out[:, 0] = np.clip(out[:, 0], 0, obj.imW - 1)
        out[:, 1] = np.clip(out[:, 1], 0, obj.imH - 1)
        out[:, 2] = np.clip(out[:, 2], 0, obj.imW - 1)
        out[:, 3] = np.clip(out[:, 3], 0, obj.imH - 1)
    return out


def select_block(aboxes, keep, num, conf):
    """Kept rows -> (block [B, nms_topN_post + 1, 14], counts [B] int32): rows [0, count) of image b are its detections,
    the rest zero, row nms_topN_post carries the count -- the all-gather message of m3dssd_amd.dist."""
    L = _hip.lib()
    dev = aboxes.device
    B, n = aboxes.shape[0], aboxes.shape[1]
    post = int(conf.nms_topN_post)
    with torch.cuda.device(dev):
        block = torch.empty(B, post + 1, 14, device=dev, dtype=torch.float32)
        counts = torch.empty(B, device=dev, dtype=torch.int32)
        _hip.check(L.m3d_select_post(aboxes.data_ptr(), keep.data_ptr(), num.data_ptr(), B, n, post, block.data_ptr(),
                                     counts.data_ptr(), _stream(dev)))
    return block, counts


def select_post(aboxes, keep, num, conf):
    """Kept rows -> fixed-size blocks (dets [B, nms_topN_post, 14] zero-padded, counts [B] int32)."""
    block, counts = select_block(aboxes, keep, num, conf)
    return block[:, :-1], counts


def detect_batch(net, im, conf, scale=None):
    """-> (dets [B, nms_topN_post, 14] zero-padded, counts [B] int32) device tensors (dets is a view of the
    [B, nms_topN_post + 1, 14] gather block, see select_block).  scale: see detect_device."""
    return select_post(*detect_device(net, im, conf, scale=scale), conf)
