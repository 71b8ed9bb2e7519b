"""RPN module + ``build`` with the reference's interface (model/M3d_inference_align.py:31-331).

``RPN.forward(x)`` returns exactly what the reference does in eval mode --
``(cls [B,N,4], prob [B,N,4], bbox_2d [B,N,4], bbox_3d [B,N,7], feat_size [2], rois [N,5])`` --
computed by the HIP engine.  Training (phase='train') is out of scope for this path."""
import os

import numpy as np
import torch
from torch import nn

from .. import rpn_util
from ..engine import Engine
from .align import center_align, shape_align
from .attention import ANAB
from .dla import DLASeg

BOX_HEADS_A = ["bbox_x", "bbox_y", "bbox_w", "bbox_h", "bbox_x3d", "bbox_y3d"]
BOX_HEADS_B = ["bbox_w3d", "bbox_h3d", "bbox_l3d", "bbox_rY3d"]


def _head(cin, mid, cout, k0):
    return nn.Sequential(nn.Conv2d(cin, mid, k0, padding=k0 // 2), nn.BatchNorm2d(mid), nn.LeakyReLU(inplace=True),
                         nn.Conv2d(mid, mid, 1), nn.BatchNorm2d(mid), nn.LeakyReLU(inplace=True),
                         nn.Conv2d(mid, cout, 1))


def _invalidate_after_load(module, incompatible_keys):
    module.refresh_engine()


class RPN(nn.Module):
    def __init__(self, phase, base, conf):
        super().__init__()
        self.base = base
        self.phase = phase
        self.device = conf.device
        self.num_classes = len(conf["lbls"]) + 1
        self.num_anchors = conf["anchors"].shape[0]
        self.anchors = torch.tensor(conf.anchors, dtype=torch.float, device=self.device, requires_grad=False)
        self.bbox_means, self.bbox_stds = conf.bbox_means[0], conf.bbox_stds[0]
        self.base_channels, self.head_channels = self.base.out_channels, 256
        self.back_bone, self.batch_size = conf.back_bone, conf.batch_size
        self.align_type = conf.align_type if "align_type" in conf else "max"
        self.attention = conf.attention if "attention" in conf else None
        self.feat_stride = conf.feat_stride
        self.feat_size = rpn_util.calc_output_size(np.array(conf.crop_size), self.feat_stride)
        self.rois = rpn_util.locate_anchors(conf.anchors, self.feat_size, conf.feat_stride, convert_tensor=True)
        self.rois = self.rois.float().to(self.device)
        from .detect import check_conf
        check_conf(conf)
        if not (conf.center_align and conf.shape_align and self.attention == "ANAB"):
            raise NotImplementedError("this build implements the anab_fullalign configuration "
                                      "(center_align, shape_align, attention='ANAB')")
        c, m, a = self.base_channels, self.head_channels, self.num_anchors
        self.cls = _head(c, m, a * self.num_classes, 3)
        for h in BOX_HEADS_A:
            setattr(self, h, _head(c, m, a, 1))
        self.center_align2d = center_align(c, self.anchors, xy_mean=self.bbox_means[0:2], xy_std=self.bbox_stds[0:2],
                                           feat_stride=self.feat_stride, feat_size=self.feat_size, kernel_size=1, k=1,
                                           thresh=0.5)
        self.center_align3d = center_align(c, self.anchors, xy_mean=self.bbox_means[4:6], xy_std=self.bbox_stds[4:6],
                                           feat_stride=self.feat_stride, feat_size=self.feat_size, kernel_size=1, k=1,
                                           thresh=0.5)
        self.shape_align = shape_align(c, self.anchors, feat_stride=self.feat_stride, feat_size=self.feat_size,
                                       kernel_size=3, k=1, thresh=0.5)
        self.bbox_z3d = _head(c, m, a, 1)
        self.bbox_z3d_gl = nn.Sequential(ANAB(c, 1), nn.BatchNorm2d(c), nn.LeakyReLU(inplace=True))
        for h in BOX_HEADS_B:
            setattr(self, h, _head(c, m, a, 1))
        self.softmax = nn.Softmax(dim=1)
        self._conf = conf
        self._engine = None
        self._param_sig = None
        # a parent's load_state_dict reaches sub-modules through _load_from_state_dict only: invalidate from there
        self.register_load_state_dict_post_hook(_invalidate_after_load)
        self.compute_dtype = str(conf.compute_dtype) if "compute_dtype" in conf else "f32"
        self.reuse_outputs = bool(conf.reuse_outputs) if "reuse_outputs" in conf else False

    # -- engine management: parameters are folded / packed once and re-packed when they change --------------------------------
    # Everything that replaces or moves parameters through the nn.Module API marks the packed engine stale:
    #   * load_state_dict on this module OR on any parent / wrapper (nn.DataParallel, DDP, a container): the parent recurses
    #     through child._load_from_state_dict and never calls the child's load_state_dict, so the invalidation hangs on
    #     _load_from_state_dict of every sub-module via a load_state_dict post-hook registered in __init__;
    #   * .to / .cuda / .float (through _apply).
    # Code that writes parameter storage IN PLACE (p.data.copy_, optimiser steps) calls refresh_engine() itself (on the SOURCE
    # module: it also drops the per-device engines that DataParallel replicas use) -- hashing
    # ~540 (data_ptr, _version) pairs on every forward cost 0.3 ms of host time per call; M3D_CHECK_PARAMS=1 turns that full
    # check on for debugging (forward then raises on a stale packing instead of using it).
    def refresh_engine(self):
        self._engine = None
        self._param_sig = None
        self.__dict__.pop("_device_engines", None)
        return self

    def _apply(self, fn, *args, **kwargs):
        self.refresh_engine()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, state_dict, *args, **kwargs):
        """Accepts the checkpoints the reference writes: they are saved from the nn.DataParallel wrapper
        (scripts/test_rpn_3d.py:50-54) and carry a leading 'module.' on every key, which the reference strips in
        lib/core.py:489-499 (load_weights(remove_module=True)); the same stripping happens here when EVERY key has it."""
        self.refresh_engine()
        keys = list(state_dict.keys())
        if keys and all(k.startswith("module.") for k in keys):
            meta = getattr(state_dict, "_metadata", None)
            state_dict = type(state_dict)((k[len("module."):], v) for k, v in state_dict.items())
            if meta is not None:                       # per-module versions: 'module' -> '', 'module.base' -> 'base'
                state_dict._metadata = type(meta)(("" if k == "module" else k[len("module."):], v)
                                                  for k, v in meta.items() if k == "module" or k.startswith("module."))
        return super().load_state_dict(state_dict, *args, **kwargs)

    def _replicate_for_data_parallel(self):
        """nn.DataParallel (the reference's scripts/test_rpn_3d.py:50-51) replicates the module per device on EVERY forward.  A
        real replica has no parameters (torch.nn.parallel.replicate hangs the broadcast copies on it as plain attributes and
        leaves `_parameters` empty), so it cannot pack an engine of its own: it asks the SOURCE module, which keeps one packed
        engine per device (`_engine_for`) built from its own state_dict moved there -- packed once, not per forward.  With ONE
        visible device DataParallel calls the module itself and nothing is replicated.  For throughput use one process per GPU
        (m3dssd_amd.dist) instead."""
        replica = super()._replicate_for_data_parallel()
        replica._engine = None
        replica._param_sig = None
        replica._is_replica = True
        object.__setattr__(replica, "_src", getattr(self, "_src", None) or self)      # (not registered as a sub-module)
        return replica

    def _engine_for(self, dev):
        """The packed engine of this (source) module on device `dev`: its own for the device its parameters live on, else a
        per-device engine packed from the state_dict moved to `dev` (cached; dropped by refresh_engine / load_state_dict / .to)."""
        dev = torch.device(dev)
        own = next(self.parameters()).device
        if dev == own:
            return self.engine()
        cache = self.__dict__.setdefault("_device_engines", {})
        key = (str(dev), self.compute_dtype)
        check = os.environ.get("M3D_CHECK_PARAMS", "0") == "1"
        if key in cache and check and cache[key][1] != self._signature():
            # the reference's DataParallel re-broadcasts the parameters on every forward; a replica engine packed from an older
            # state of the source's parameters would silently keep using it
            raise RuntimeError("RPN: parameters changed in place since the replica engine on %s packed them; call "
                               "net.refresh_engine() on the source module" % (dev,))
        if key not in cache:
            if dev.type != "cuda":
                raise NotImplementedError("RPN.forward runs on a ROCm device only (replica on %s)" % (dev,))
            sd = {k: v.to(dev) for k, v in self.state_dict().items()}
            if self.compute_dtype == "bf16":
                from ..engine_bf16 import EngineBF16
                eng = EngineBF16(sd, self._conf, device=dev)
            else:
                eng = Engine(sd, self._conf, device=dev)
            cache[key] = (eng, self._signature() if check else None)
        return cache[key][0]

    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters()) + \
            tuple((b.data_ptr(), b._version) for b in self.buffers())

    def set_compute_dtype(self, dtype):
        """'f32' (default: the reference's arithmetic) or 'bf16' (bf16 storage / MFMA, fp32 accumulation; see engine_bf16.py)."""
        if str(dtype) != self.compute_dtype:
            self.compute_dtype = str(dtype)
            self._engine = None
        return self

    def engine(self, device=None):
        if getattr(self, "_is_replica", False):     # a DataParallel replica: the source module owns the per-device engines
            if device is None:
                held = [t for t in self.__dict__.values() if torch.is_tensor(t) and t.is_cuda]
                former = list(getattr(self, "_former_parameters", {}).values())
                for m in self.modules():
                    former += list(getattr(m, "_former_parameters", {}).values())
                    if former:
                        break
                device = (former or held)[0].device if (former or held) else None
            if device is None:
                raise RuntimeError("RPN replica: cannot tell its device (no replicated parameters); call engine(device)")
            return self._src._engine_for(device)
        if self._engine is not None and os.environ.get("M3D_CHECK_PARAMS", "0") == "1" and self._param_sig != self._signature():
            raise RuntimeError("RPN: parameters changed in place since the engine packed them; call net.refresh_engine()")
        if self._engine is None:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise NotImplementedError("RPN.forward runs on a ROCm device only; move the module with .to('cuda') "
                                          "(the reference's DCNv2 has no CPU path either, dcn_v2_func.py:23-24)")
            if self.compute_dtype == "bf16":      # BASELINE.json configs[2]: bf16 storage + bf16 MFMA, fp32 accumulation
                from ..engine_bf16 import EngineBF16
                self._engine = EngineBF16(self.state_dict(), self._conf, device=dev)
            elif self.compute_dtype in ("f32", "fp32", "float32"):
                self._engine = Engine(self.state_dict(), self._conf, device=dev)
            else:
                raise ValueError("compute_dtype must be 'f32' or 'bf16' (got %r)" % (self.compute_dtype,))
            self._param_sig = self._signature() if os.environ.get("M3D_CHECK_PARAMS", "0") == "1" else None
        return self._engine

    def forward(self, x):
        """Eval-mode forward of the reference (M3d_inference_align.py:303-313).  The four big outputs are FRESH tensors, as the
        reference's are: a caller may keep them across iterations.  ``conf.reuse_outputs = True`` (or ``net.reuse_outputs = True``)
        returns views of the engine's plan-owned buffers instead, which the next forward of the same shape overwrites -- the form
        the device detection stage (lib.rpn_util.detect_batch / im_detect_3d, PipelinedDetector) uses internally: it consumes the
        outputs before the next forward.  Fresh outputs cost nothing but their allocation: `m3d_bundle_outputs` writes them in place
        (round 6; no copies)."""
        return self._forward_views(x, fresh=not self.reuse_outputs)

    def _forward_views(self, x, fresh=False):
        if self.training:
            raise NotImplementedError("m3dssd_amd accelerates inference (eval mode); call .eval() or build(conf, 'test')")
        u8 = x.dtype == torch.uint8 and x.dim() == 4 and x.shape[3] == 3      # raw BGR frames [B, h, w, 3]: the test-time
        if u8:                                                                 # Preprocess runs inside the stem kernel
            size = [int(v) for v in self._conf.crop_size]
            feat_h, feat_w = size[0] // self.feat_stride, size[1] // self.feat_stride
        else:
            feat_h, feat_w = x.shape[2] // self.feat_stride, x.shape[3] // self.feat_stride
            assert feat_h == self.feat_size[0], "x.shape is {}".format(x.shape)
        with torch.no_grad():
            if u8:
                cls, prob, bbox_2d, bbox_3d = self.engine(x.device).forward_u8(x, size, fresh=fresh)
            else:
                cls, prob, bbox_2d, bbox_3d = self.engine(x.device).forward(x.float(), fresh=fresh)
        key = (feat_h, feat_w, x.device)
        if getattr(self, "_feat_size_key", None) != key:       # cached: a fresh host->device copy per call would
            self._feat_size_t = torch.tensor([feat_h, feat_w], dtype=torch.float, device=x.device)  # break graph capture
            self._feat_size_key = key
        feat_size = self._feat_size_t
        if self.feat_size[0] != feat_h or self.feat_size[1] != feat_w:
            self.feat_size = [feat_h, feat_w]
            self.rois = rpn_util.locate_anchors(self.anchors, self.feat_size, self.feat_stride, convert_tensor=True)
            self.rois = self.rois.float().to(x.device)
        if self.rois.device != x.device:
            self.rois = self.rois.to(x.device)
        return cls, prob, bbox_2d, bbox_3d, feat_size, self.rois


def build(conf, phase="train"):
    train = phase.lower() == "train"
    base_name = conf.back_bone
    if base_name[0:3] != "dla":
        raise NotImplementedError
    base = DLASeg(base_name, pretrained=conf.pre_train, down_ratio=conf.feat_stride, final_kernel=1, last_level=5,
                  head_conv=256, conf=conf)
    rpn_net = RPN(phase, base, conf)
    if train:
        rpn_net.train()
    else:
        rpn_net.eval()
    return rpn_net
