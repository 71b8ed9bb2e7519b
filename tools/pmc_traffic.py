#!/usr/bin/env python
"""HBM traffic per launch from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

Follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes read -> doubled here; WRITE_SIZE is taken as is.  Round 4 calibrated both on known byte
counts per ACCESS WIDTH (tools/ubench/pmc_calib.hip, profiles/r04_pmc_calibration.txt: 1 GiB through a buffer 4x the
Infinity Cache): FETCH_SIZE x 2.000 for 4 / 8 / 16 B per lane, for 64-byte segments (the F(4x4) patch loads), for the
8-lanes-per-line gather map and for LDS-DMA loads alike; WRITE_SIZE x 1.000 for 4 / 16 B per lane and 64-byte segments --
one correction serves every kernel.  (FETCH_SIZE still counts Infinity-Cache hits: it is fabric traffic, an upper bound of HBM.)
usage: python tools/pmc_traffic.py gpurun_out/pmc3 profiles/r01_hbm_traffic.json [launches.json]

launches.json (``bench.py --dump-launches``: the ordered [family label, kernel base name] list of the MFMA launches of ONE
step) turns the per-SYMBOL averages into per-FAMILY ones: one kernel symbol serves several engine families (e.g.
conv_wave_kernel<true,4> = the full-size deformable layers AND their split-K variants on small maps), so the rows of the
counter CSV (dispatch order) are aligned with the repeating per-step launch sequence and averaged per family label.
"""
import collections
import csv
import json
import os
import sys


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            n[r["Kernel_Name"]] += 1
    return {k: tot[k] / n[k] for k in tot}, n


def base_name(kernel_name):
    n = kernel_name.split("(", 1)[0].split("<", 1)[0].strip()
    return n[5:] if n.startswith("void ") else n


def per_family(path, counter, launches):
    """Align the CSV's MFMA-kernel rows (dispatch order) with the repeating per-step sequence `launches`."""
    bases = set(b for _, b in launches)
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and base_name(r["Kernel_Name"]) in bases]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    n = len(launches)
    if not rows or len(rows) % n:
        raise SystemExit("%s: %d MFMA-kernel rows are not a multiple of the %d launches of a step" % (path, len(rows), n))
    tot, cnt, grids = collections.defaultdict(float), collections.defaultdict(int), collections.defaultdict(set)
    for i, r in enumerate(rows):
        label, b = launches[i % n]
        if base_name(r["Kernel_Name"]) != b:
            raise SystemExit("%s: dispatch %s is %s, the step sequence expects %s (%s) at position %d"
                             % (path, r["Dispatch_Id"], base_name(r["Kernel_Name"]), b, label, i % n))
        tot[label] += float(r["Counter_Value"])
        cnt[label] += 1
        grids[label].add((r["Kernel_Name"], int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)))
    return {k: tot[k] / cnt[k] for k in tot}, cnt, grids


def main(d, out, launches=None):
    f, nf = per_kernel(os.path.join(d, "FETCH_SIZE_counter_collection.csv"), "FETCH_SIZE")
    w, nw = per_kernel(os.path.join(d, "WRITE_SIZE_counter_collection.csv"), "WRITE_SIZE")
    fam = {}
    if launches:
        seq = [tuple(x) for x in json.load(open(launches))]
        ff, nff, grids = per_family(os.path.join(d, "FETCH_SIZE_counter_collection.csv"), "FETCH_SIZE", seq)
        wf, _, _ = per_family(os.path.join(d, "WRITE_SIZE_counter_collection.csv"), "WRITE_SIZE", seq)
        for k in ff:
            fam[k] = {"launches": nff[k], "fetch_kib_raw": round(ff[k], 1), "write_kib": round(wf[k], 1),
                      "hbm_bytes_per_launch": int((2.0 * ff[k] + wf[k]) * 1024),
                      "kernels_and_workgroups": sorted("%s x %d" % g for g in grids[k])}
    res = {}
    for k in f:
        if k in w:
            res[k] = {"launches": nf[k], "fetch_kib_raw": round(f[k], 1), "write_kib": round(w[k], 1),
                      "hbm_bytes_per_launch": int((2.0 * f[k] + w[k]) * 1024)}
    csrc_files = None
    try:                               # the library that just ran under rocprofv3 (this script runs on the GPU box right behind the passes)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from m3dssd_amd import _hip
        csrc_files = _hip.lib_source_hashes()
    except Exception as e:             # noqa: BLE001
        print("pmc_traffic: no source record of the library (%s): bench.py will report this pass as stale" % (e,))
    json.dump({"csrc_files": csrc_files,
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs of "
                         "`bench.py --steps 3 --warmup 2 --no-graph`; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB "
                         "(factors calibrated per access width in profiles/r04_pmc_calibration.txt: 2.000 / 1.000 for every width); `families` = the same rows averaged per "
                         "engine family label (aligned with bench.py --dump-launches by dispatch order)",
               "families": fam, "kernels": res},
              open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]:
        print("%-70s n=%4d  %8.1f MB/launch" % (k[:70], v["launches"], v["hbm_bytes_per_launch"] / 1e6))
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
        print("family %-62s n=%4d  %8.1f MB/launch" % (k[:62], v["launches"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
