"""CPU tests of bench.py's host logic that needs no GPU: the kernel-symbol grouping behind the headline roofline and the
self-check of the committed PMC traffic figures against the library that is loaded (VERDICT r4 #3a / #6)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from m3dssd_amd import _hip  # noqa: E402


def test_loaded_library_was_built_from_the_sources_in_the_tree():
    """m3d_source_hashes() (compiled into the library by the Makefile) == sha256 of the csrc / include files next to it: a stale
    .so would make every `traffic_stale` answer -- and every GPU test -- speak about other code than the tree's."""
    lib, tree = _hip.lib_source_hashes(), _hip.tree_source_hashes()
    assert lib == tree, {k: (lib.get(k), tree.get(k)) for k in set(lib) | set(tree) if lib.get(k) != tree.get(k)}


def test_traffic_is_reported_stale_when_the_kernel_source_changed(tmp_path):
    lib = _hip.lib_source_hashes()
    fam = {"conv_wave<deform,4>": {"hbm_bytes_per_launch": 81234567}, "bf16_head_mlp": {"hbm_bytes_per_launch": 5}}
    json.dump({"csrc_files": dict(lib), "families": fam}, open(tmp_path / "r99_f32_hbm_traffic.json", "w"))
    tr, src, stale = bench.pmc_traffic("conv_wave<deform,4>", lib, str(tmp_path))
    assert tr == 81234567 and src.endswith("#families") and stale is False
    # the deformable kernel's source differs from the profiled build: its families are stale, an unrelated family is not
    other = dict(lib, **{"dcn_wave.hip": "0" * 16})
    json.dump({"csrc_files": other, "families": fam}, open(tmp_path / "r99_f32_hbm_traffic.json", "w"))
    assert bench.pmc_traffic("conv_wave<deform,4>", lib, str(tmp_path))[2] is True
    assert bench.pmc_traffic("bf16_head_mlp", lib, str(tmp_path))[2] is False
    # a shared header changed: everything is stale; a pass without a record (rounds 1-4) cannot be checked: stale
    json.dump({"csrc_files": dict(lib, **{"common.h": "1" * 16}), "families": fam}, open(tmp_path / "r99_f32_hbm_traffic.json", "w"))
    assert bench.pmc_traffic("bf16_head_mlp", lib, str(tmp_path))[2] is True
    json.dump({"families": fam}, open(tmp_path / "r99_f32_hbm_traffic.json", "w"))
    assert bench.pmc_traffic("bf16_head_mlp", lib, str(tmp_path))[2] is True
    assert bench.pmc_traffic("wino44<16,16>", lib, str(tmp_path)) == (None, None, None)


def test_families_of_one_symbol_are_grouped_for_the_headline_roofline():
    labels = ["conv_wave<deform,4>", "conv_wave<deform,4,splitk2>", "conv_wave<deform,4,splitk4>", "conv_wave<plain,4>",
              "head_mlp<3,64>", "head_mlp<2,192>", "wino44<16,16>", "wino44<16,16,kpair>", "wino44<16,16,splitk4>"]
    sym = {}
    for k in labels:
        sym.setdefault(bench.kernel_symbol(k), []).append(k)
    assert sym["void conv_wave_kernel<true, 4>(ConvWaveArgs)"] == labels[:3]
    assert sym["void conv_wave_kernel<false, 4>(ConvWaveArgs)"] == [labels[3]]
    assert len(sym["void head_mlp_kernel<true, 64>(MlpBatch)"]) == 1
    assert bench.kernel_symbol("wino44<16,16,kpair>") != bench.kernel_symbol("wino44<16,16>")
    for k in labels:                                         # every family maps to sources that exist in the library's record
        assert set(bench.family_sources(k)) <= set(_hip.lib_source_hashes())
