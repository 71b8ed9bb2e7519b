"""m3dssd_amd -- MI355X (gfx950) native inference path for M3DSSD.

Hot path only: DLA-34 backbone, DCNv2 alignment stages, ANAB
attention, RPN heads, decode and NMS, implemented as hand-written HIP kernels behind a
C ABI (include/m3dssd_hip.h, m3dssd_amd/csrc) and exposed through the reference's own
module / function signatures (model/*, lib/* shims at the repo root).
"""
__version__ = "0.1.0"
