"""Drop-in for the reference's model/module/attention.py (ANAB, PAPAModule)."""
from m3dssd_amd.host.attention import ANAB, PAPAModule  # noqa: F401
