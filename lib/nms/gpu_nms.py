"""Drop-in for the reference's Cython lib/nms/gpu_nms.pyx."""
from m3dssd_amd.host.nms import gpu_nms  # noqa: F401
