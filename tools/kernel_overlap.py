#!/usr/bin/env python
"""Do the kernels of a rocprofv3 --kernel-trace run overlap in time?  Reads the rocpd SQLite database and prints, for the
busiest stretch of the trace, the sum of the kernel durations against the length of the union of their [start, end] intervals
(ratio 1.0 = strictly one kernel at a time), plus the pairs of kernel names that overlap most.
usage: python tools/kernel_overlap.py gpurun_out/prof/x_results.db [max_rows]
"""
import collections
import sqlite3
import sys


def main(db, limit=200000):
    c = sqlite3.connect(db)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    src = None
    for cand in ["kernels"] + [n for n in names if "kernel" in n.lower()]:
        if cand not in names:
            continue
        cols = [r[1] for r in c.execute("pragma table_info(%s)" % cand)]
        if "start" in cols and "end" in cols and "name" in cols:
            src = cand
            break
    if src is None:
        print("no kernel table with start/end/name; tables:", names)
        return
    rows = c.execute("select name, start, end from %s order by start limit %d" % (src, limit)).fetchall()
    print("%s: %d dispatches" % (src, len(rows)))
    rows = rows[len(rows) // 2:]                    # steady state: second half of the trace
    tot = sum(e - s for _, s, e in rows)
    union, cur_s, cur_e = 0, None, None
    pairs = collections.Counter()
    active = []
    for n, s, e in rows:
        active = [(an, ae) for an, ae in active if ae > s]
        for an, ae in active:
            pairs[(an[:60], n[:60])] += min(ae, e) - s
        active.append((n, e))
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    span = rows[-1][2] - rows[0][1]
    print("sum of durations %.3f ms, union %.3f ms, ratio %.3f, span %.3f ms (idle %.1f %%)" % (
        tot / 1e6, union / 1e6, tot / union, span / 1e6, 100.0 * (1 - union / span)))
    for (a, b), ns in pairs.most_common(12):
        print("  %8.3f ms  %s || %s" % (ns / 1e6, a, b))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 200000)
