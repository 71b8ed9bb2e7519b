"""Drop-in for the reference's model/pose_dla_dcn.py (DLA-34 path)."""
from m3dssd_amd.host.dla import (BasicBlock, Root, Tree, DLA, dla34, DeformConv, IDAUp, DLAUp, DLASeg,  # noqa: F401
                                 fill_up_weights, BN_MOMENTUM)
from m3dssd_amd.host.dcn import DCN  # noqa: F401
