"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/m3dssd_hip.h declares (no compute calls -- there is no GPU here), the ctypes signature table
covers exactly the header, and the import-path shims expose the reference's names."""
import os
import re

import pytest
import torch

from m3dssd_amd import _hip


def _declared():
    hdr = open(_hip.HEADER).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(m3d_[a-z0-9_]+|_nms)\s*\(", hdr))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_hip.SO_PATH), "run __graft_entry__.build() first"
    L = _hip.lib()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), n
    assert names == set(_hip.SIGNATURES), (names ^ set(_hip.SIGNATURES))
    assert L.m3d_abi_version() == 5


def test_conv_desc_layout_matches_header():
    """Field order/count of the ctypes mirror follows the C struct (names differ only for `in`)."""
    hdr = open(_hip.HEADER).read()
    body = hdr[hdr.index("typedef struct m3d_conv_desc {"):hdr.index("} m3d_conv_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";")[:-1]:
        decl = decl.split("{")[-1].strip()
        if not decl:
            continue
        names = decl.replace("*", " ").split()
        first = len([t for t in names if t in ("const", "float", "int", "long")])
        fields += [n.strip(",") for n in names[first:]]
    mine = [("in" if n == "inp" else n) for n, _ in _hip.ConvDesc._fields_]
    assert fields == mine


def test_shims_expose_reference_names():
    import lib.nms.gpu_nms as g
    import lib.rpn_util as r
    import model.DCNv2.dcn_v2 as d
    import model.M3d_inference_align as m
    import model.module.attention as a
    import model.module.feturealign_mgpu as f
    import model.pose_dla_dcn as p
    for mod, names in ((m, ["build", "RPN"]), (p, ["DLASeg", "DeformConv", "DLA", "dla34", "IDAUp", "DLAUp"]),
                       (d, ["DCNv2", "DCN"]), (a, ["ANAB", "PAPAModule"]), (f, ["center_align", "shape_align"]),
                       (g, ["gpu_nms"]), (r, ["locate_anchors", "calc_output_size", "flatten_tensor", "im_detect_3d"])):
        for n in names:
            assert hasattr(mod, n), (mod.__name__, n)


def test_product_path_fails_loudly_without_a_gpu():
    """No CPU fallback anywhere on the product path."""
    from m3dssd_amd import synth
    from model.M3d_inference_align import build
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    conf = synth.synth_conf((128, 320), 0, device="cpu")
    net = build(conf, "test")
    with pytest.raises(NotImplementedError):
        net(torch.zeros(1, 3, 128, 320))
    import model.DCNv2.dcn_v2 as d
    with pytest.raises(NotImplementedError):
        d.DCNv2(4, 4, 3, 1, 1)(torch.zeros(1, 4, 4, 4), torch.zeros(1, 18, 4, 4), torch.zeros(1, 9, 4, 4))


def test_product_does_not_import_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for base in ("m3dssd_amd", "model", "lib"):
        for dp, _, files in os.walk(os.path.join(root, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    txt = open(os.path.join(dp, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "liboracle" in txt:
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
