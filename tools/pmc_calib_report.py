#!/usr/bin/env python
"""Calibration factors of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters per access pattern (tools/ubench/pmc_calib.hip).

usage: python tools/pmc_calib_report.py DIR out.json      DIR holds FETCH_SIZE/ and WRITE_SIZE/ rocprofv3 csv outputs
Every calib_* kernel moves exactly 1 GiB through a buffer four times the Infinity Cache; factor = bytes moved / (counter KiB *
1024): the multiplier that turns the counter into bytes for that access pattern (the guide's "x2" for 16 B/lane reads)."""
import collections
import csv
import glob
import json
import os
import sys

BYTES = float(1 << 30)


def rows(d, counter):
    f = glob.glob(os.path.join(d, counter, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        raise SystemExit("no %s counter csv under %s" % (counter, d))
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("calib_"):
            out[r["Kernel_Name"].split("(", 1)[0]].append(float(r["Counter_Value"]))
    return out


def main(d, out):
    fetch, write = rows(d, "FETCH_SIZE"), rows(d, "WRITE_SIZE")
    res = {}
    print("%-26s %14s %14s %10s %10s" % ("kernel (1 GiB moved)", "FETCH_SIZE KiB", "WRITE_SIZE KiB", "fetch x", "write x"))
    for k in sorted(set(fetch) | set(write)):
        fv = sorted(fetch.get(k, [0.0]))[len(fetch.get(k, [0.0])) // 2]
        wv = sorted(write.get(k, [0.0]))[len(write.get(k, [0.0])) // 2]
        is_load = "_load_" in k
        ff = BYTES / (fv * 1024) if (is_load and fv > 0) else None
        wf = BYTES / (wv * 1024) if (not is_load and wv > 0) else None
        res[k] = {"fetch_kib": fv, "write_kib": wv, "fetch_factor": ff, "write_factor": wf}
        print("%-26s %14.0f %14.0f %10s %10s" % (k, fv, wv, "%.3f" % ff if ff else "-", "%.3f" % wf if wf else "-"))
    json.dump({"method": "tools/ubench/pmc_calib.hip: each kernel moves 1 GiB once through a 1 GiB buffer (4x the Infinity Cache); "
                         "factor = bytes / (counter KiB * 1024), median of 3 launches; separate rocprofv3 --pmc passes",
               "kernels": res}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
