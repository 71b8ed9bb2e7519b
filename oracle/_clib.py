"""ORACLE (test infrastructure): build + load oracle/build/liboracle.so via ctypes."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "liboracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("dcn_im2col.c", "nms_ref.c")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int)
        L.oracle_dcn_im2col.argtypes = [fp, fp, fp] + [ctypes.c_int] * 14 + [fp]
        L.oracle_dcn_im2col.restype = None
        L.oracle_nms_sorted.argtypes = [ip, ip, fp, ctypes.c_int, ctypes.c_int, ctypes.c_float]
        L.oracle_nms_sorted.restype = None
        L.oracle_iou.argtypes = [fp, fp]
        L.oracle_iou.restype = ctypes.c_float
        _lib = L
    return _lib
