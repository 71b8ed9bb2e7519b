#!/usr/bin/env python
"""HBM traffic per launch from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

Follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled here (the igemm and
MLP loaders are 16 B/lane); WRITE_SIZE is taken as is (uncalibrated per the guide -- ratios are reliable).
usage: python tools/pmc_traffic.py gpurun_out/pmc3 profiles/r01_hbm_traffic.json
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


def main(d, out):
    f, nf = per_kernel(os.path.join(d, "FETCH_SIZE_counter_collection.csv"), "FETCH_SIZE")
    w, nw = per_kernel(os.path.join(d, "WRITE_SIZE_counter_collection.csv"), "WRITE_SIZE")
    res = {}
    for k in f:
        if k in w:
            res[k] = {"launches": nf[k], "fetch_kib_raw": round(f[k], 1), "write_kib": round(w[k], 1),
                      "hbm_bytes_per_launch": int((2.0 * f[k] + w[k]) * 1024)}
    json.dump({"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs of "
                         "`bench.py --steps 3 --warmup 2 --no-graph`; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB "
                         "(gfx950 wide-read correction of MI355X_MICROARCH.md)", "kernels": res},
              open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]:
        print("%-70s n=%4d  %8.1f MB/launch" % (k[:70], v["launches"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
