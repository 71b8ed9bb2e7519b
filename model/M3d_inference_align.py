"""Drop-in for the reference's model/M3d_inference_align.py (``build``, ``RPN``)."""
from m3dssd_amd.host.rpn import RPN, build  # noqa: F401
from m3dssd_amd.host.dla import DLASeg, DeformConv  # noqa: F401
from m3dssd_amd.host.dcn import DCNv2  # noqa: F401
from m3dssd_amd.host.attention import ANAB  # noqa: F401
from m3dssd_amd.host.align import shape_align, center_align  # noqa: F401
