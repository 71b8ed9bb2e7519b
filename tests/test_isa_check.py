"""CPU tests of the static ISA check (tools/check_isa_hazards.py): the scanner must flag the store-data write-after-read pattern that
round 4 found in compiler output (a VALU write of a 16-byte store's data registers inside the hazard window) and spills of the
kernels with hand-counted vmcnt, and must accept the padded forms."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import check_isa_hazards as chk  # noqa: E402

BAD = """
_Z13wino44_kernelILi1ELi2ELi2EEv10Wino44Args:
	buffer_store_dwordx4 v[4:7], v48, s[28:31], s0 offen
	v_pk_fma_f32 v[4:5], v[38:39], v[32:33], v[42:43]
	s_endpgm
"""
GOOD = """
_Z13wino44_kernelILi1ELi2ELi2EEv10Wino44Args:
	;;#ASMSTART
	buffer_store_dwordx4 v[4:7], v48, s[28:31], s0 offen
	s_nop 1
	;;#ASMEND
	v_pk_fma_f32 v[4:5], v[38:39], v[32:33], v[42:43]
	buffer_store_dwordx4 v[8:11], v48, s[28:31], s0 offen
	v_pk_fma_f32 v[12:13], v[38:39], v[32:33], v[42:43]
	buffer_store_dwordx2 v[4:5], v48, s[28:31], s0 offen
	v_mov_b32_e32 v4, 0
	s_endpgm
"""


def test_store_data_hazard_is_flagged_and_padded_form_passes():
    f, _ = chk.scan(BAD)
    assert len(f) == 1 and f[0][0].startswith("_Z13wino44_kernel") and f[0][4] == 0 and f[0][5] == 1
    assert chk.scan(GOOD)[0] == []


def test_global_and_literal_soffset_stores_need_two_wait_states():
    one = "k:\n\tglobal_store_dwordx4 v[16:17], v[0:3], off offset:16\n\ts_waitcnt lgkmcnt(0)\n\tv_mul_u32_u24_e32 v1, 0x888, v13\n"
    two = "k:\n\tglobal_store_dwordx4 v[16:17], v[0:3], off offset:16\n\ts_nop 1\n\tv_mul_u32_u24_e32 v1, 0x888, v13\n"
    lit = "k:\n\tbuffer_store_dwordx4 v[0:3], v9, s[0:3], 0 offen\n\ts_nop 0\n\tv_mov_b32_e32 v2, 0\n"
    addr = "k:\n\tglobal_store_dwordx4 v[16:17], v[0:3], off\n\tv_mov_b32_e32 v16, 0\n"        # address registers are safe
    assert len(chk.scan(one)[0]) == 1 and chk.scan(two)[0] == [] and len(chk.scan(lit)[0]) == 1 and chk.scan(addr)[0] == []


def test_scratch_in_a_hand_counted_kernel_is_an_error():
    meta = """
amdhsa.kernels:
  - .name:           _Z13wino44_kernelILi1ELi2ELi1EEv10Wino44Args
    .private_segment_fixed_size: 20
    .symbol:         _Z13wino44_kernelILi1ELi2ELi1EEv10Wino44Args.kd
  - .name:           _Z15refine3d_kernel10RefineArgs
    .private_segment_fixed_size: 496
    .symbol:         _Z15refine3d_kernel10RefineArgs.kd
  - .name:           _Z16head_mlp_kernelILb1ELi64EEv8MlpBatch
    .private_segment_fixed_size: 0
    .symbol:         _Z16head_mlp_kernelILb1ELi64EEv8MlpBatch.kd
"""
    _, spills = chk.scan(meta)
    assert spills == [("_Z13wino44_kernelILi1ELi2ELi1EEv10Wino44Args", 20), ("_Z15refine3d_kernel10RefineArgs", 496)]
    assert any(h in spills[0][0] for h in chk.HAND_COUNTED) and not any(h in spills[1][0] for h in chk.HAND_COUNTED)


def test_back_edge_and_swap_destinations_are_followed():
    """a 16-byte global store at the end of a loop body whose back edge leads to a write of its data registers (one wait
    state: the branch) must be flagged; so must a v_swap that writes a data register as its SECOND operand."""
    loop = ("k:\n.LBB0_1:\n\tv_mov_b32_e32 v2, 0\n\tv_add_u32_e32 v9, 1, v9\n\tglobal_store_dwordx4 v[16:17], v[0:3], off\n"
            "\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n")
    f, _ = chk.scan(loop)
    assert len(f) == 1 and f[0][4] == 1 and f[0][5] == 2 and "v_mov_b32_e32 v2" in f[0][3]
    padded = loop.replace("\ts_cbranch_scc1", "\ts_nop 1\n\ts_cbranch_scc1")
    assert chk.scan(padded)[0] == []
    uncond = ("k:\n\tglobal_store_dwordx4 v[16:17], v[0:3], off\n\ts_branch .LBB0_2\n\tv_mov_b32_e32 v1, 0\n.LBB0_2:\n"
              "\tv_mov_b32_e32 v9, 0\n\ts_endpgm\n")
    assert chk.scan(uncond)[0] == []                    # the write behind an unconditional branch is not on the path
    swap = "k:\n\tglobal_store_dwordx4 v[16:17], v[0:3], off\n\tv_swap_b32 v20, v3\n\ts_endpgm\n"
    assert len(chk.scan(swap)[0]) == 1
