"""Drop-in for the reference's model/DCNv2/dcn_v2_func.py (DCNv2Function, forward only)."""
from m3dssd_amd.host.dcn import DCNv2Function  # noqa: F401
