"""GPU tests of the detection stage behind the forward (-m gpu): the planar form that the pipelined
detector runs (sort keys from the planar class logits, top-N-pre rows decoded from the planar staging the heads write) against the
bundled form (m3d_bundle_outputs + m3d_topk_decode_scaled) that is itself pinned to the reference's im_detect_3d golden rows in
tests/test_gpu_parity.py.  Integer / row identity: bit for bit."""
import ctypes

import numpy as np
import pytest
import torch

from m3dssd_amd import synth

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("no ROCm device visible: the gpu-marked tests must run on the MI355X box")
    return torch.device("cuda:0")


def _net(crop, B, dtype="f32"):
    from model.M3d_inference_align import build
    conf = synth.synth_conf(crop, 0, batch_size=B, device="cuda:0")
    net = build(conf, "test")
    net.load_state_dict(synth.synth_state_dict(0))
    return net.to(_dev()).set_compute_dtype(dtype), conf


@pytest.mark.parametrize("crop,B", [((128, 320), 3), ((384, 1280), 2)])
def test_planar_keys_and_decode_equal_bundled_form(crop, B):
    """m3d_score_keys_planar == the score_bits of m3d_bundle_outputs, and m3d_topk_decode_planar == m3d_topk_decode_scaled on the
    bundled tensors (with and without test-time scale factors), bit for bit, rows_out included."""
    from m3dssd_amd import _hip
    from m3dssd_amd.host.detect import detect_from_outputs, detect_from_planar, score_keys_planar
    dev = _dev()
    L = _hip.lib()
    net, conf = _net(crop, B)
    x = synth.synth_frames(B, crop, 77).to(dev)
    with torch.no_grad():
        cls, prob, b2, b3, _, rois = net(x)
    eng = net.engine()
    plan = eng.plan_for(B, crop[0], crop[1])
    bits_bundle = plan.named["score_bits"].clone()
    plan.named["score_bits"].zero_()
    score_keys_planar(eng, plan)
    assert torch.equal(plan.named["score_bits"], bits_bundle)
    for scale in (None, torch.tensor([1.0, 0.75, 1.3][:B], device=dev)):
        a0, k0, n0 = detect_from_outputs(eng, plan, prob, b2, b3, rois, conf, scale)
        a1, k1, n1 = detect_from_planar(eng, plan, rois, conf, scale)
        assert torch.equal(a0, a1) and torch.equal(n0, n1)
        for b in range(B):
            assert torch.equal(k0[b, :int(n0[b])], k1[b, :int(n1[b])])
    # selected row ids through the C ABI
    R = prob.shape[1]
    k = min(int(conf.nms_topN_pre), R)
    ws = torch.empty(L.m3d_topk_decode_workspace_bytes(B, R), device=dev, dtype=torch.uint8)
    rows0 = torch.empty(B, k, device=dev, dtype=torch.int32)
    rows1 = torch.empty(B, k, device=dev, dtype=torch.int32)
    ab = torch.empty(B, k, 14, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = eng.P
    _hip.check(L.m3d_topk_decode(bits_bundle.data_ptr(), prob.data_ptr(), b2.data_ptr(), b3.data_ptr(), rois.data_ptr(),
                                 P["anchors"].data_ptr(), P["means"].data_ptr(), P["stds"].data_ptr(), ab.data_ptr(),
                                 rows0.data_ptr(), ws.data_ptr(), ws.numel(), B, R, k, st))
    _hip.check(L.m3d_topk_decode_planar(bits_bundle.data_ptr(), plan.named["cls_planar"].data_ptr(),
                                        plan.named["box_planar"].data_ptr(), rois.data_ptr(), P["anchors"].data_ptr(),
                                        P["means"].data_ptr(), P["stds"].data_ptr(), None, ab.data_ptr(), rows1.data_ptr(),
                                        ws.data_ptr(), ws.numel(), B, eng.A, R // eng.A, k, st))
    assert torch.equal(rows0, rows1)
    # argument checks: misaligned / ragged HW, k out of range, short workspace
    assert L.m3d_score_keys_planar(plan.named["cls_planar"].data_ptr(), bits_bundle.data_ptr(), B, eng.A, 6, st) != 0
    assert L.m3d_topk_decode_planar(bits_bundle.data_ptr(), plan.named["cls_planar"].data_ptr(),
                                    plan.named["box_planar"].data_ptr(), rois.data_ptr(), P["anchors"].data_ptr(),
                                    P["means"].data_ptr(), P["stds"].data_ptr(), None, ab.data_ptr(), None, ws.data_ptr(), 16,
                                    B, eng.A, R // eng.A, k, st) != 0
    assert b"workspace" in L.m3d_last_error()


@pytest.mark.parametrize("planar", [True, False])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_pipelined_detector_planar_and_bundled_forms_equal_detect_batch(planar, dtype):
    """One hipGraph per step = forward(k) beside detect(k-1): both forms of the detector (key-only pass + planar decode; bundled
    tensors) return the rows of the sequential detect_batch, bit for bit, for every batch of a sequence."""
    from lib.rpn_util import detect_batch
    from m3dssd_amd.pipeline import PipelinedDetector
    dev = _dev()
    net, conf = _net((128, 320), 2, dtype)
    xs = [synth.synth_frames(2, (128, 320), 40 + i).to(dev) for i in range(4)]
    ref = []
    for x in xs:
        d, c = detect_batch(net, x, conf)
        ref.append((d.clone(), c.clone()))
    pipe = PipelinedDetector(net, conf, 2, 128, 320, planar=planar)
    assert pipe.planar is planar and (pipe.n_join < pipe.n_fwd) == planar
    got = []
    for x in xs:
        r = pipe.step(x)
        if r is not None:
            got.append((r[0].clone(), r[1].clone()))
    r = pipe.flush()
    got.append((r[0].clone(), r[1].clone()))
    assert len(got) == len(ref)
    for (gd, gc), (rd, rc) in zip(got, ref):
        assert torch.equal(gc, rc) and torch.equal(gd, rd)
        assert int(gc.sum()) > 0
