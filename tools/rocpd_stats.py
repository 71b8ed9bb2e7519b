#!/usr/bin/env python
"""Export the per-kernel summary (rocprofv3 --kernel-trace --stats) from a rocpd results .db to CSV.

ROCm 7.2's rocprofv3 writes a rocpd SQLite database by default; its `top_kernels` view is the
--stats kernel summary (name, calls, total / average duration in us, percentage).
usage: python tools/rocpd_stats.py gpurun_out/prof/r01_results.db profiles/r01_bench_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for name, calls, tot, avg, pct in rows:
            w.writerow([name if len(name) < 200 else name[:197] + "...", calls, "%.3f" % tot, "%.3f" % avg, "%.4f" % pct])
    print("wrote %d kernels to %s" % (len(rows), out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
