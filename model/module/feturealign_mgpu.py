"""Drop-in for the reference's model/module/feturealign_mgpu.py (center_align, shape_align)."""
from m3dssd_amd.host.align import center_align, shape_align  # noqa: F401
