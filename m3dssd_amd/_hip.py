"""ctypes binding of libm3dssd_hip.so (include/m3dssd_hip.h).

The library is built in-tree (m3dssd_amd/csrc/build/) by ``build()`` -- plain hipcc for gfx950,
no torch headers.  There is NO fallback: if the shared object is missing or a symbol is absent the
import fails loudly, and every entry point that returns a non-zero status raises RuntimeError with
the library's message (the reference surfaced THError/THArgCheck the same way).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.environ.get("M3D_HIP_LIB") or os.path.join(CSRC, "build", "libm3dssd_hip.so")   # (override: diagnostic builds only)
HEADER = os.path.join(os.path.dirname(_HERE), "include", "m3dssd_hip.h")

c_int, c_float, c_ll, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_longlong, ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    """Mirror of ``m3d_conv_desc``."""
    _fields_ = [
        ("inp", c_void_p), ("in_cs", c_int),
        ("N", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int),
        ("wgt", c_void_p), ("wgt_img_stride", c_ll),
        ("Cout", c_int), ("Cout_pad", c_int),
        ("kh", c_int), ("kw", c_int), ("stride", c_int), ("pad", c_int), ("dil", c_int),
        ("Ho", c_int), ("Wo", c_int),
        ("out", c_void_p), ("out_cs", c_int), ("out_nchw", c_int), ("out_img_stride", c_ll),
        ("scale", c_void_p), ("shift", c_void_p),
        ("res", c_void_p), ("res_cs", c_int), ("res_mode", c_int),
        ("act", c_int), ("sigmoid_from", c_int),
        ("dcn_offmask", c_void_p), ("dcn_om_cs", c_int),
        ("splitk_ws", c_void_p), ("splitk_ws_bytes", c_ll),
    ]


class MlpDesc(ctypes.Structure):
    """Mirror of ``m3d_mlp_desc``."""
    _fields_ = [
        ("inp", c_void_p), ("in_cs", c_int), ("M", c_ll), ("Cin", c_int),
        ("w1", c_void_p), ("s1", c_void_p), ("t1", c_void_p),
        ("w2", c_void_p), ("s2", c_void_p), ("t2", c_void_p),
        ("w3", c_void_p), ("s3", c_void_p), ("t3", c_void_p),
        ("Cout", c_int), ("Cout_pad", c_int),
        ("out", c_void_p), ("out_img_stride", c_ll), ("HW", c_int),
    ]


class ConvBf16Desc(ctypes.Structure):
    """Mirror of ``m3d_conv_bf16_desc``."""
    _fields_ = [
        ("inp", c_void_p), ("in_cs", c_int),
        ("N", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int),
        ("wgt", c_void_p), ("wgt_img_stride", c_ll),
        ("Cout", c_int), ("Cout_pad", c_int), ("Kpad", c_int),
        ("kh", c_int), ("kw", c_int), ("stride", c_int), ("pad", c_int),
        ("Ho", c_int), ("Wo", c_int),
        ("out", c_void_p), ("out_cs", c_int), ("out_mode", c_int), ("out_img_stride", c_ll),
        ("scale", c_void_p), ("shift", c_void_p),
        ("res", c_void_p), ("res_cs", c_int), ("res_mode", c_int),
        ("act", c_int), ("sigmoid_from", c_int),
        ("dcn_offmask", c_void_p), ("dcn_om_cs", c_int),
        ("groups", c_int),
        ("in_group_off", c_ll), ("wgt_group_off", c_ll), ("out_group_off", c_ll),
        ("ss_group_off", c_int),
        ("wgt_f16", c_void_p), ("dcn_ws", c_void_p), ("dcn_ws_bytes", c_ll),
        ("wgt_wave", c_void_p),
    ]


class HeadBf16Desc(ctypes.Structure):
    """Mirror of ``m3d_head_bf16_desc``."""
    _fields_ = [
        ("inp", c_void_p), ("in_cs", c_int), ("M", c_ll), ("Cin", c_int),
        ("w1", c_void_p), ("w2", c_void_p), ("w3", c_void_p),
        ("s1", c_void_p), ("t1", c_void_p), ("s2", c_void_p), ("t2", c_void_p), ("s3", c_void_p), ("t3", c_void_p),
        ("Cout", c_int), ("Cout_pad", c_int),
        ("out", c_void_p), ("out_group_off", c_ll), ("out_img_stride", c_ll),
        ("HW", c_int), ("groups", c_int),
    ]


class Head2Bf16Desc(ctypes.Structure):
    """Mirror of ``m3d_head2_bf16_desc``."""
    _fields_ = [
        ("inp", c_void_p), ("in_cs", c_int), ("M", c_ll),
        ("w1f", c_void_p), ("w2f", c_void_p), ("w3", c_void_p),
        ("t1", c_void_p), ("t2", c_void_p), ("t3", c_void_p),
        ("Cout", c_int),
        ("out", c_void_p), ("out_group_off", c_ll), ("out_img_stride", c_ll),
        ("HW", c_int), ("groups", c_int),
    ]


class Tail2Bf16Desc(ctypes.Structure):
    """Mirror of ``m3d_tail2_bf16_desc``."""
    _fields_ = [
        ("inp", c_void_p), ("in_cs", c_int), ("M", c_ll),
        ("waf", c_void_p), ("wbf", c_void_p), ("t1", c_void_p), ("t2", c_void_p),
        ("Cout", c_int),
        ("out", c_void_p), ("out_img_stride", c_ll),
        ("HW", c_int),
    ]


class TreeEntryBf16Desc(ctypes.Structure):
    """Mirror of ``m3d_tree_entry_bf16_desc``."""
    _fields_ = [
        ("inp", c_void_p), ("in_cs", c_int), ("N", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int), ("Cout", c_int),
        ("wfrag", c_void_p), ("shift1", c_void_p), ("shiftp", c_void_p),
        ("t", c_void_p), ("t_cs", c_int), ("res", c_void_p), ("res_cs", c_int), ("bottom", c_void_p), ("bottom_cs", c_int),
    ]


class QkvsBf16Desc(ctypes.Structure):
    """Mirror of ``m3d_qkvs_bf16_desc``."""
    _fields_ = [
        ("inp", c_void_p), ("in_cs", c_int), ("M", c_ll), ("wf", c_void_p),
        ("q", c_void_p), ("q_cs", c_int), ("q_rows", c_int),
        ("kv", c_void_p), ("kv_cs", c_int), ("kv_rows", c_int),
        ("s", c_void_p), ("s_cs", c_int), ("s_rows", c_int),
    ]


P = c_void_p
# name -> (restype, argtypes); every name here must be declared in include/m3dssd_hip.h
SIGNATURES = {
    "m3d_last_error": (ctypes.c_char_p, []),
    "m3d_abi_version": (c_int, []),
    "m3d_source_hashes": (ctypes.c_char_p, []),
    "m3d_conv2d_forward": (c_int, [ctypes.POINTER(ConvDesc), P]),
    "m3d_conv_bf16_forward": (c_int, [ctypes.POINTER(ConvBf16Desc), P]),
    "m3d_conv_bf16_dcn_ws_bytes": (c_ll, [c_int, c_int, c_int]),
    "m3d_conv_bf16_variant": (c_int, [ctypes.POINTER(ConvBf16Desc)]),
    "m3d_head_mlp_bf16_forward": (c_int, [ctypes.POINTER(HeadBf16Desc), P]),
    "m3d_head_mlp2_bf16_forward": (c_int, [ctypes.POINTER(Head2Bf16Desc), P]),
    "m3d_tree_entry_bf16_applicable": (c_int, [ctypes.POINTER(TreeEntryBf16Desc)]),
    "m3d_tree_entry_bf16_forward": (c_int, [ctypes.POINTER(TreeEntryBf16Desc), P]),
    "m3d_anab_qkvs_bf16_forward": (c_int, [ctypes.POINTER(QkvsBf16Desc), P]),
    "m3d_head_tail2_bf16_forward": (c_int, [ctypes.POINTER(Tail2Bf16Desc), P]),
    "m3d_stem_conv7x7_bf16": (c_int, [P, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                      P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "m3d_frontend2_bf16_forward": (c_int, [P, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
                                   + [P] * 7 + [c_int] * 4 + [P]),
    "m3d_anab_attend_bf16": (c_int, [P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, c_int, P, c_int, P]),
    "m3d_maxpool2x2_bf16": (c_int, [P, c_int, P, c_int] + [c_int] * 4 + [P]),
    "m3d_upsample2x_add_bf16": (c_int, [P, c_int, P, P, c_int, P, c_int] + [c_int] * 4 + [P]),
    "m3d_f32_to_bf16": (c_int, [P, P, c_ll, P]),
    "m3d_softmax_rows_bf16": (c_int, [P, c_int, c_int, c_int, P, c_int, P]),
    "m3d_head_mlp_forward": (c_int, [ctypes.POINTER(MlpDesc), P]),
    "m3d_head_mlp_forward_batched": (c_int, [ctypes.POINTER(MlpDesc), c_int, P]),
    "m3d_wino_conv3x3_forward": (c_int, [ctypes.POINTER(ConvDesc), P]),
    "m3d_wino_conv3x3_forward_ex": (c_int, [ctypes.POINTER(ConvDesc), c_int, P]),
    "m3d_wino_conv3x3_splitk_plan": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(c_int), ctypes.POINTER(c_ll)]),
    "m3d_wino_conv3x3_variant": (c_int, [ctypes.POINTER(ConvDesc)]),
    "m3d_wino44_applicable": (c_int, [ctypes.POINTER(ConvDesc)]),
    "m3d_wino44_conv3x3_forward": (c_int, [ctypes.POINTER(ConvDesc), P]),
    "m3d_wino44_conv3x3_forward_ex": (c_int, [ctypes.POINTER(ConvDesc), c_int, P]),
    "m3d_wino44_kpair": (c_int, [ctypes.POINTER(ConvDesc)]),
    "m3d_wino44_conv3x3_forward_touch": (c_int, [ctypes.POINTER(ConvDesc), c_int, P, c_ll, P]),
    "m3d_cache_touch": (c_int, [P, c_ll, P]),
    "m3d_upload_indirect": (c_int, [P, P, c_ll, P]),
    "m3d_wino44_splitk_plan": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(c_int), ctypes.POINTER(c_ll)]),
    "m3d_conv_wave_applicable": (c_int, [ctypes.POINTER(ConvDesc)]),
    "m3d_conv_wave_splitk_plan": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(c_int), ctypes.POINTER(c_ll)]),
    "m3d_conv_wave_forward": (c_int, [ctypes.POINTER(ConvDesc), P]),
    "m3d_conv2d_tile": (c_int, [ctypes.POINTER(ConvDesc)] + [ctypes.POINTER(c_int)] * 4),
    "m3d_conv2d_splitk_plan": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(c_int), ctypes.POINTER(c_ll)]),
    "m3d_dcn_v2_workspace_bytes": (c_ll, [c_int] * 10),
    "m3d_dcn_v2_workspace_bytes_grouped": (c_ll, [c_int] * 11),
    "m3d_dcn_v2_forward": (c_int, [P] * 6 + [c_int] * 14 + [P, c_ll, P]),
    "m3d_pack_conv_weight": (c_int, [P, P] + [c_int] * 6 + [P]),
    "m3d_nchw_to_nhwc": (c_int, [P, P] + [c_int] * 5 + [P]),
    "m3d_nhwc_to_nchw": (c_int, [P, c_int, P] + [c_int] * 4 + [P]),
    "m3d_stem_conv7x7": (c_int, [P, P, P, P, P] + [c_int] * 4 + [P]),
    "m3d_refine_3d": (c_int, [P, P, c_int, c_int, P, P, ctypes.c_double, c_int, ctypes.c_double, ctypes.c_double, P, P]),
    "m3d_refine_3d_ex": (c_int, [P, P, c_int, c_int, P, P, P, P, ctypes.c_double, c_int, ctypes.c_double, ctypes.c_double, P, P]),
    "m3d_preprocess_u8": (c_int, [P, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), P, c_int,
                                  c_int, P]),
    "m3d_stem_conv7x7_u8": (c_int, [P, c_int, c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), P, P, P, P,
                                    c_int, c_int, c_int, c_int, P]),
    "m3d_conv3x3_c16": (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "m3d_conv3x3_c16_wino": (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "m3d_maxpool2x2": (c_int, [P, c_int, P, c_int] + [c_int] * 4 + [P]),
    "m3d_upsample2x_add": (c_int, [P, c_int, P, P, c_int, P, c_int] + [c_int] * 4 + [P]),
    "m3d_anchor_select": (c_int, [P] + [c_int] * 4 + [P, P, P, P]),
    "m3d_anchor_select_keys": (c_int, [P] + [c_int] * 3 + [P, P, P, P]),
    "m3d_fg_top1": (c_int, [P, c_int, c_int, c_int, P, P, P]),
    "m3d_align_offsets": (c_int, [c_int, P, P, c_float, P, P, P, P] + [c_float] * 4 + [P] + [c_int] * 5 + [c_ll, P]),
    "m3d_anab_pool_partial": (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, P] + [c_int] * 5 + [P]),
    "m3d_anab_pool_finish": (c_int, [P, P, P] + [c_int] * 4 + [P, c_int, c_int, P, c_int, c_int, P]),
    "m3d_anab_pool_nested_scratch_bytes": (c_ll, [c_int, c_int]),
    "m3d_anab_pool_nested": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int, P, c_int, P]),
    "m3d_anab_attend_f32": (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, P, P, c_int, P, c_int, P]),
    "m3d_anab_pool_nested_bf16": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int, P, c_int, P]),
    "m3d_anab_pool_nested_bf16_ex": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int, P, c_int, P, P, P]),
    "m3d_softmax_rows": (c_int, [P, c_int, c_int, c_int, P]),
    "m3d_bundle_outputs": (c_int, [P] * 7 + [c_int] * 3 + [P]),
    "m3d_decode_rows": (c_int, [P] * 9 + [c_int] * 3 + [P]),
    "m3d_topk_decode_workspace_bytes": (c_ll, [c_int, c_int]),
    "m3d_topk_decode": (c_int, [P] * 11 + [c_ll] + [c_int] * 3 + [P]),
    "m3d_topk_decode_scaled": (c_int, [P] * 12 + [c_ll] + [c_int] * 3 + [P]),
    "m3d_score_keys_planar": (c_int, [P, P, c_int, c_int, c_int, P]),
    "m3d_topk_decode_planar": (c_int, [P] * 11 + [c_ll] + [c_int] * 4 + [P]),
    "m3d_select_post": (c_int, [P] * 3 + [c_int] * 3 + [P, P, P]),
    "m3d_nms_workspace_bytes": (c_ll, [c_int, c_int]),
    "m3d_nms_sorted_dev": (c_int, [P, c_int, c_int, c_int, c_float, P, P, P, P]),
    "_nms": (None, [P, P, P, c_int, c_int, c_float, c_int]),
    "m3d_rotate_iou_eval": (c_int, [P, c_int, P, c_int, c_int, P, P]),
    "m3d_eval_image_box_overlap": (c_int, [P, c_int, P, c_int, c_int, P]),
    "m3d_eval_d3_overlap": (c_int, [P, c_int, P, c_int, P, c_int]),
    "m3d_eval_statistics": (c_int, [P, c_ll, P, c_int, P, c_int, P, P, P, c_int, c_int, ctypes.c_double, ctypes.c_double,
                                    c_int, c_int, P, P, ctypes.POINTER(c_int)]),
    "m3d_eval_fused_statistics": (c_int, [P, c_ll, P, P, P, P, c_int, P, P, P, P, P, c_int, ctypes.c_double, P, c_int, c_int]),
    "m3d_clock_probe": (c_int, [P, ctypes.c_double, P]),
    "m3d_event_create": (c_int, [ctypes.POINTER(c_void_p)]),
    "m3d_event_record": (c_int, [P, P]),
    "m3d_event_elapsed_ms": (c_int, [P, P, ctypes.POINTER(c_float)]),
    "m3d_event_destroy": (c_int, [P]),
}

_lib = None


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into csrc/build/libm3dssd_hip.so (make, hipcc)."""
    cmd = ["make", "-C", CSRC, "-j", str(os.cpu_count() or 4)] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libm3dssd_hip.so failed")
    return SO_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "libm3dssd_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                "there is no non-HIP fallback for the M3DSSD hot path" % SO_PATH)
        # torch first: PyTorch-ROCm ships its own libamdhip64; a library dlopen()ed BEFORE torch binds the system's /opt/rocm copy and
        # the process ends up with two HIP runtimes -- kernels registered in one, the device initialised in the other ("no
        # ROCm-capable device is detected" at the first launch; seen with `python __graft_entry__.py smoke` = build() then smoke()
        # in one process, round 6).  With torch loaded the library's libamdhip64 dependency resolves to the copy torch already mapped.
        import torch  # noqa: F401
        L = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)            # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def lib_source_hashes():
    """{file name: sha256[:16]} of the sources the LOADED library was built from (m3d_source_hashes)."""
    txt = lib().m3d_source_hashes().decode()
    return dict(item.split(":", 1) for item in txt.split(";") if item)


def tree_source_hashes():
    """The same record computed from the sources in the tree (what the library would carry if it were rebuilt now)."""
    import hashlib
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") or f.endswith(".h")] + [HEADER]
    return {os.path.basename(f): hashlib.sha256(open(f, "rb").read()).hexdigest()[:16] for f in files}


def check(status):
    if status != 0:
        raise RuntimeError("m3dssd_hip error %d: %s" % (status, lib().m3d_last_error().decode()))
