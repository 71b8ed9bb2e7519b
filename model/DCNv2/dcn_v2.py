"""Drop-in for the reference's model/DCNv2/dcn_v2.py (DCNv2, DCN)."""
from m3dssd_amd.host.dcn import DCNv2, DCN, DCNv2Function  # noqa: F401
