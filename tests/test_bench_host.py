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
    assert sym["void conv_wave_kernel<true, 4, 4, 2>(ConvWaveArgs)"] == labels[:3]
    assert sym["void conv_wave_kernel<false, 4, 4, 3>(ConvWaveArgs)"] == [labels[3]]
    assert len(sym["void head_mlp_kernel<true, 64>(MlpBatch)"]) == 1
    assert bench.kernel_symbol("wino44<16,16,kpair>") != bench.kernel_symbol("wino44<16,16>")
    for k in labels:                                         # every family maps to sources that exist in the library's record
        assert set(bench.family_sources(k)) <= set(_hip.lib_source_hashes())


def _canned_record():
    """A full bench record of the size round 5 printed (24 KB: 18 + 16 kernel families, helper tables, notes)."""
    fam = {"fam<%d,with,a,long,label>" % i: {"kernel": "void some_kernel_template<%d, true, false, 128>(SomeArgsStruct)" % i,
                                             "launches_per_step": 3, "ms_per_step": 0.5, "executed_tflops": 90.1,
                                             "frac_of_mfma_peak": 0.57, "algorithmic_bytes_per_launch": 57642356,
                                             "algorithmic_gbs": 457.1, "hbm_frac": 0.057, "nearer_roof": "mfma", "traffic": 71751793,
                                             "traffic_ratio": 1.24, "traffic_source": "profiles/r06_f32_hbm_traffic.json#families",
                                             "traffic_stale": False} for i in range(20)}
    roof = {"bound": "mfma", "kernel": "void conv_wave_kernel<true, 4>(ConvWaveArgs)", "achieved": 94.33, "peak": 157.3,
            "unit": "TFLOP/s", "frac": 0.5997, "note": "n" * 300, "traffic": 71751793, "traffic_source": "x" * 60,
            "traffic_stale": False, "traffic_unit": "u" * 250, "families_of_symbol": list(fam)[:4],
            "algorithmic_bytes_per_launch": 57642356, "launches_timed": 11, "avg_launch_ms": 0.1261, "share_of_gpu_time": 0.225}
    one = {"metric": "images/sec at 1280x384 bs=8, 1/2/4/8 MI355X; 3D-box Linf vs ref", "value": 1349.62, "unit": "images/sec",
           "n_gpus": 1, "steps": 200, "warmup": 3, "ms_per_step": 5.928, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "w" * 400, "per_gpu_batch": 8, "global_batch": 8, "resolution": [1280, 384], "parallelism": "dp1"},
           "launch": "l" * 300, "work_in_timed_region": "forward + decode + top-3000 + NMS + select; bundle_outputs NOT run",
           "roofline": roof, "gpu_ms_by_kernel_one_step": {k: 0.5 for k in fam}, "mfma_kernel_families": fam,
           "mfma_time_weighted_frac": 0.543, "helper_kernels": {("helper%d" % i): dict(fam["fam<0,with,a,long,label>"]) for i in range(10)},
           "sclk_under_step_ghz": 2.338, "roofline_frac_at_held_clock": 0.6157,
           "step_roofline": {"algorithmic_tflops": 142.8, "mfma_frac": 0.9078, "hbm_frac": 0.1756, "note": "n" * 400},
           "dropin": {"steps": 60, "eager": {"value": 1100.0, "ms_per_step": 7.27}, "graph": {"value": 1290.0, "ms_per_step": 6.2},
                      "vs_headline_graph": 0.956}}
    rec = dict(one)
    rec["configs2_bf16"] = dict(one, dtype="bf16", config=dict(one["config"], per_gpu_batch=64))
    rec["configs3_shard32"] = {"workload": "w" * 100, "f32": {"value": 1496.3, "ms_per_step": 21.4, "steps": 30, "note": "n" * 200},
                               "bf16": {"value": 5676.6, "ms_per_step": 5.6, "steps": 30}, "allgather_us_one_rank_rccl": 15.4,
                               "allgather_bytes_per_rank": 73472}
    rec["feed_u8"] = {"value": 1321.1, "ms_per_step": 6.056, "vs_resident": 0.98, "input": "i" * 300, "h2d_bytes_per_step": 11178000}
    rec["cpu_baseline"] = {"value": 4.997, "unit": "images/sec", "cores": 16, "host_cores": 256, "kind": "port",
                           "cpu_model": "AMD EPYC 9575F 64-Core Processor", "sample": "s" * 400}
    return rec


def test_stdout_line_is_under_4_kb_and_keeps_the_contract_keys(tmp_path):
    """the driver keeps only the tail of stdout; round 5's 24 KB line left BENCH_r05.json with parsed = null.
    The line built from a record of that size is < 4096 bytes, one line, and holds the contract keys + roofline + cpu_baseline +
    the compact side legs; the full record goes to the detail file."""
    rec = _canned_record()
    assert len(json.dumps(rec)) > 20000
    line = bench.compact_line(rec, "bench_detail.json")
    assert len(line) < 4096 and "\n" not in line
    r = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["vs_baseline"] is None and r["config"]["workload"] and "model" not in r["config"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "traffic_ratio")) <= set(r["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(r["cpu_baseline"])
    assert r["configs2_bf16"]["value"] == rec["configs2_bf16"]["value"] and r["configs2_bf16"]["roofline"]["frac"] == 0.5997
    assert r["configs2_bf16"]["per_gpu_batch"] == 64 and r["configs2_bf16"]["step"]["mfma_frac"] == 0.9078
    assert r["dropin"]["graph"]["value"] == 1290.0 and r["configs2_bf16"]["dropin"]["eager"]["ms_per_step"] == 7.27
    assert r["configs3_shard32"]["f32"]["value"] == 1496.3 and r["feed_u8"]["value"] == 1321.1
    assert "mfma_kernel_families" not in r and "helper_kernels" not in r
    # emit(): detail file written, ONE line on the given descriptor
    rd, wr = os.pipe()
    bench.emit(rec, str(tmp_path / "d.json"), wr)
    os.close(wr)
    got = os.read(rd, 1 << 16).decode()
    os.close(rd)
    assert got.count("\n") == 1 and len(got) < 4096 and json.loads(got)["detail"] == str(tmp_path / "d.json")
    assert json.load(open(tmp_path / "d.json"))["mfma_kernel_families"] == rec["mfma_kernel_families"]


def test_stdout_line_sheds_optional_blocks_rather_than_growing():
    rec = _canned_record()
    rec["cpu_baseline"]["cpu_model"] = "c" * 1500          # something unexpected grows: optional blocks go, the headline stays
    rec["config"]["parallelism"] = "p" * 1200
    r = json.loads(bench.compact_line(rec, "bench_detail.json"))
    assert len(json.dumps(r, separators=(",", ":"))) < 4096
    assert r["value"] == 1349.62 and "roofline" in r and "cpu_baseline" in r
