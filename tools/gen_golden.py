#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own Python in this container.

Build-container only: needs /root/reference (read-only) and never travels to the GPU
box.  What it does:

* puts /root/reference first on sys.path and stubs the third-party imports the
  reference pulls in but the hot path never calls (torchvision, cv2, shapely, numba,
  skimage, tensorboardX, fire, easydict) plus the two native extensions that cannot be
  built here: ``lib.nms.gpu_nms`` (-> the reference's own lib/nms/py_cpu_nms.py) and
  ``model.DCNv2.dcn_v2_func`` (-> oracle/dcn.py, the CPU restatement of
  dcn_v2_cuda_forward; the op has no CPU implementation in the reference);
* builds ``model.M3d_inference_align.build(conf, 'test')`` with back_bone='dla34',
  loads m3dssd_amd.synth.synth_state_dict with strict=True (this is the state_dict
  contract check: 542 keys, identical names and shapes), runs forward on seeded frames;
* dumps reference outputs: full for a 128x320 crop, strided samples + float64 checksums
  for 384x1280; the reference's locate_anchors / generate_anchors / flatten_tensor;
  lib/nms/py_cpu_nms keep lists; im_detect_3d rows (torch .cuda() shimmed to a no-op).

Run:  python tools/gen_golden.py        (about a minute)
"""
import importlib
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")


def _install_stubs():
    # The repo ships import-path shims named ``model`` / ``lib`` (regular packages), which would win over
    # the reference's namespace packages: import what we need from the repo first, then drop it from sys.path.
    sys.path.insert(0, REPO)
    import m3dssd_amd.synth  # noqa: F401
    import oracle.dcn  # noqa: F401
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
    for k in [k for k in sys.modules if k == "model" or k.startswith("model.") or k == "lib" or k.startswith("lib.")]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    for name in ["torchvision", "torchvision.models", "cv2", "shapely", "shapely.geometry", "numba",
                 "numba.cuda", "skimage", "skimage.io", "tensorboardX", "fire", "mpl_toolkits",
                 "mpl_toolkits.mplot3d"]:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = MagicMock()
    ed = types.ModuleType("easydict")

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed

    # native extension 1: NMS -> the reference's own pure-python NMS
    spec = importlib.util.spec_from_file_location("ref_py_cpu_nms", os.path.join(REF, "lib/nms/py_cpu_nms.py"))
    pynms = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pynms)
    g = types.ModuleType("lib.nms.gpu_nms")
    g.gpu_nms = lambda dets, thresh, device_id=0: pynms.py_cpu_nms(dets, thresh)
    sys.modules["lib.nms.gpu_nms"] = g

    # native extension 2: DCNv2 -> oracle restatement (legacy instance-style call convention,
    # model/DCNv2/dcn_v2_func.py:13-38)
    from oracle import dcn as odcn
    f = types.ModuleType("model.DCNv2.dcn_v2_func")

    class DCNv2Function:
        def __init__(self, stride, padding, dilation=1, deformable_groups=1):
            self.a = (stride, padding, dilation, deformable_groups)

        def __call__(self, input, offset, mask, weight, bias):
            return odcn.dcn_v2_forward(input, offset, mask, weight, bias, *self.a)

    f.DCNv2Function = DCNv2Function
    f.DCNv2PoolingFunction = MagicMock()
    sys.modules["model.DCNv2.dcn_v2_func"] = f
    return pynms


def _checks(t):
    a = t.detach().double().numpy() if hasattr(t, "detach") else np.asarray(t, dtype=np.float64)
    return np.array([a.sum(), np.abs(a).sum(), float(a.size)])


def main():
    pynms = _install_stubs()
    import torch
    torch.set_num_threads(8)
    from m3dssd_amd import synth
    import lib.rpn_util as ref_rpn
    import model.M3d_inference_align as ref_model
    from easydict import EasyDict

    os.makedirs(OUT, exist_ok=True)
    sd = synth.synth_state_dict(0)

    # ---------------- anchors / rois / flatten --------------------------------------------
    conf0 = synth.synth_conf((384, 1280), 0, device="cpu")
    rc = EasyDict(dict(anchor_scales=conf0.anchor_scales, anchor_ratios=conf0.anchor_ratios,
                       feat_stride=8, cluster_anchors=0, has_3d=False))  # 3-D stats need the KITTI imdb
    ref_rpn.generate_anchors(rc, None, None)
    rois_small = ref_rpn.locate_anchors(conf0.anchors, [4, 6], 8, convert_tensor=True)
    rois_full = ref_rpn.locate_anchors(conf0.anchors, [48, 160], 8, convert_tensor=True).float()
    rois_np = ref_rpn.locate_anchors(conf0.anchors, [4, 6], 8, convert_tensor=False)
    t = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).view(2, 3, 4, 5)
    np.savez_compressed(
        os.path.join(OUT, "anchors.npz"), anchors_2d=rc.anchors, rois_4x6=rois_small.numpy(),
        rois_4x6_np=rois_np, rois_full_rows=rois_full[::997].numpy(), rois_full_chk=_checks(rois_full),
        out_size=ref_rpn.calc_output_size(np.array([384, 1280]), 8),
        out_size_odd=ref_rpn.calc_output_size(np.array([370, 1225]), 8),
        flat_in=t.numpy(), flat_out=ref_rpn.flatten_tensor(t).numpy())

    # ---------------- NMS: reference py_cpu_nms ---------------------------------------------
    nms = {}
    for n in (1, 2, 63, 64, 65, 300, 3000):
        d = synth.synth_boxes(n, seed=n)
        nms["dets_%d" % n] = d
        nms["keep_%d" % n] = np.asarray(pynms.py_cpu_nms(d, 0.4), dtype=np.int64)
    d = synth.synth_boxes(300, seed=7)
    d[:, :4] = np.round(d[:, :4] / 16) * 16           # many exactly-equal coordinates / IoU ties at thresholds
    nms["dets_grid"] = d
    for thr in (0.0, 0.25, 0.4, 0.5, 1.0):
        nms["keep_grid_%g" % thr] = np.asarray(pynms.py_cpu_nms(d, thr), dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "nms.npz"), **nms)

    # ---------------- model forward ---------------------------------------------------------
    def run(crop, batch, pad):
        conf = synth.synth_conf(crop, 0, batch_size=batch, device="cpu")
        rconf = EasyDict(dict(conf))
        net = ref_model.build(rconf, "test")
        missing = net.load_state_dict(sd, strict=True)
        ref_keys = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        my_keys = [(k, tuple(v.shape)) for k, v in sd.items()]
        assert ref_keys == my_keys, "state_dict contract mismatch"
        x = synth.synth_frames(batch, crop, 1234, pad_right_third=pad)
        taps = {}
        hooks = []
        want = {"base.base.level2": "level2", "base.base.level5": "level5", "base": "feats0",
                "shape_align": "feats", "center_align2d": "feats_align2d",
                "center_align3d": "feats_align3d", "bbox_z3d_gl": "feats_gl",
                "base.dla_up.ida_0.proj_1": "base.dla_up.ida_0.proj_1.out",
                "base.ida_up.node_1": "base.ida_up.node_1.out"}
        mods = dict(net.named_modules())
        for mname, tname in want.items():
            hooks.append(mods[mname].register_forward_hook(
                lambda m, i, o, tname=tname: taps.__setitem__(tname, o.detach().clone())))
        with torch.no_grad():
            out = net(x)
        for h in hooks:
            h.remove()
        return conf, net, x, out, taps

    conf, net, x, out, taps = run((128, 320), 2, False)
    cls, prob, b2, b3, fs, rois = out
    fg = (1 - prob[:, :, 0])
    print("small: fg>0.5 fraction of rows %.3f" % (fg > 0.5).float().mean().item())
    hard = (fg.view(2, 36, 16, 40).max(dim=1)[0] > 0.5).float().mean().item()
    print("small: hard-mask (max over anchors fg > 0.5) fraction of pixels %.3f" % hard)
    rs = 4
    g = {"row_stride": np.array(rs), "cls": cls[:, ::rs].numpy(), "prob": prob[:, ::rs].numpy(),
         "bbox_2d": b2[:, ::rs].numpy(), "bbox_3d": b3[:, ::rs].numpy(),
         "feat_size": fs.numpy(), "rois": rois[::rs].numpy()}
    for name, tns in (("cls", cls), ("prob", prob), ("bbox_2d", b2), ("bbox_3d", b3), ("rois", rois)):
        g["chk." + name] = _checks(tns)
    for k, v in taps.items():
        g["tap." + k] = v[:, ::8].numpy()               # every 8th channel
        g["chk." + k] = _checks(v)
    np.savez_compressed(os.path.join(OUT, "model_128x320_b2.npz"), **g)

    # im_detect_3d on the reference outputs (batch index 0), .cuda() shimmed away
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor

    class Obj:
        imH, imW, p2, scale_factor = 128, 320, np.eye(4), 1.0

    class FakeNet:
        def eval(self):
            return self

        def __call__(self, im):
            return tuple(o.clone() for o in out)
    rconf = EasyDict(dict(conf))
    ab = ref_rpn.im_detect_3d(x[:1], FakeNet(), rconf, Obj())
    print("small: detections after NMS", ab.shape)
    np.savez_compressed(os.path.join(OUT, "detect_128x320.npz"), aboxes=ab)

    conf, net, x, out, taps = run((384, 1280), 1, True)
    cls, prob, b2, b3, fs, rois = out
    fg = (1 - prob[:, :, 0])
    print("full: fg>0.5 fraction of rows %.3f" % (fg > 0.5).float().mean().item())
    print("full: hard-mask fraction of pixels %.3f" % (fg.view(1, 36, 48, 160).max(dim=1)[0] > 0.5).float().mean().item())
    st = 211
    g = {"stride": np.array(st), "cls": cls[:, ::st].numpy(), "prob": prob[:, ::st].numpy(),
         "bbox_2d": b2[:, ::st].numpy(), "bbox_3d": b3[:, ::st].numpy(), "feat_size": fs.numpy()}
    for name, tns in (("cls", cls), ("prob", prob), ("bbox_2d", b2), ("bbox_3d", b3)):
        g["chk." + name] = _checks(tns)
    for k, v in taps.items():
        g["chk." + k] = _checks(v)
        g["tap." + k] = v[:, ::16, ::3, ::5].numpy()
    np.savez_compressed(os.path.join(OUT, "model_384x1280_b1.npz"), **g)
    ab = ref_rpn.im_detect_3d(x[:1], FakeNet(), EasyDict(dict(conf)),
                              type("O", (), dict(imH=384, imW=1280, p2=np.eye(4), scale_factor=1.0))())
    np.savez_compressed(os.path.join(OUT, "detect_384x1280.npz"), aboxes=ab)
    print("full: detections after NMS", ab.shape)
    for f in sorted(os.listdir(OUT)):
        print("%-28s %8.1f KB" % (f, os.path.getsize(os.path.join(OUT, f)) / 1024))


if __name__ == "__main__":
    main()
