#!/usr/bin/env python
"""Timeline of ONE replay of the pipelined hipGraph from a rocprofv3 kernel trace (csv): where does the step's wall time go that the
sum of the stand-alone kernel times does not explain -- idle gaps between graph nodes, or kernels stretched by what runs beside them?

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python bench.py --dtype bf16 --steps 12 --warmup 3 ...
    python tools/graph_timeline.py OUT/**/t_kernel_trace.csv [anchor-kernel-substring] [--full]

A step = the dispatches from one launch of the anchor kernel (default: the first forward kernel of the plan) to the next.  Printed for
the median-length step of the second half of the trace: length, union of the busy intervals, idle time, the sum of the durations, the
largest gaps (with the kernels either side), and per kernel name: launches, in-graph time and the time during which it ran alone.
"""
import collections
import csv
import glob
import sys


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
    rows.sort()
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    full = "--full" in sys.argv
    paths = glob.glob(args[0], recursive=True) if any(c in args[0] for c in "*?") else [args[0]]
    rows = load(paths[0])
    anchor = args[1] if len(args) > 1 else None
    if anchor is None:
        for cand in ("bf16_frontend2_kernel", "bf16_frontend_kernel", "stem_conv7x7"):
            if any(cand in r[2] for r in rows):
                anchor = cand
                break
    starts = [i for i, r in enumerate(rows) if anchor in r[2]]
    starts = starts[len(starts) // 2:]
    steps = [(rows[a][0], rows[b][0], a, b) for a, b in zip(starts[:-1], starts[1:])]
    steps.sort(key=lambda s: s[1] - s[0])
    t0, t1, a, b = steps[len(steps) // 2]
    ks = rows[a:b]
    print("anchor %s: %d steps in the second half, lengths %.3f .. %.3f ms; median step %.3f ms, %d dispatches" % (
        anchor, len(steps), (steps[0][1] - steps[0][0]) / 1e6, (steps[-1][1] - steps[-1][0]) / 1e6, (t1 - t0) / 1e6, len(ks)))
    # union / gaps
    busy, gaps, cur_e, prev = 0, [], None, None
    for s, e, n, q, st in ks:
        if cur_e is None:
            cur_s, cur_e = s, e
        elif s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, prev, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        prev = n if (cur_e == e) else prev
    busy += min(cur_e, t1) - cur_s
    tail = t1 - cur_e
    tot = sum(e - s for s, e, *_ in ks)
    print("sum of durations %.3f ms, busy (union) %.3f ms, idle inside the step %.3f ms in %d gaps (+ %.3f ms from the last end to the next anchor)" % (
        tot / 1e6, busy / 1e6, sum(g[0] for g in gaps) / 1e6, len(gaps), tail / 1e6))
    gs = sorted(g[0] for g in gaps)
    if gs:
        print("gap sizes us: min %.1f median %.1f p90 %.1f max %.1f" % (gs[0] / 1e3, gs[len(gs) // 2] / 1e3, gs[int(len(gs) * 0.9)] / 1e3, gs[-1] / 1e3))
    for g, p, n in sorted(gaps, reverse=True)[:8]:
        print("   %7.1f us  %s -> %s" % (g / 1e3, p[:50], n[:50]))
    # per name: in-graph time, alone time
    events = []
    for i, (s, e, n, q, st) in enumerate(ks):
        events.append((s, 1, i))
        events.append((e, 0, i))
    events.sort()
    active, alone, last = set(), collections.Counter(), None
    for t, kind, i in events:
        if last is not None and len(active) == 1:
            alone[ks[next(iter(active))][2]] += t - last
        if kind:
            active.add(i)
        else:
            active.discard(i)
        last = t
    per = collections.OrderedDict()
    for s, e, n, q, st in ks:
        d = per.setdefault(n, [0, 0])
        d[0] += 1
        d[1] += e - s
    print("%-70s %4s %9s %9s" % ("kernel", "n", "ms", "alone ms"))
    for n, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("%-70s %4d %9.3f %9.3f" % (n[:70], c, d / 1e6, alone[n] / 1e6))
    if full:
        print("--- dispatch list (start us, dur us, gap-to-previous-end us, queue, name)")
        pe = None
        for s, e, n, q, st in ks:
            print("%9.1f %8.1f %7.1f %s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - pe) / 1e3 if pe else 0.0, q, n[:80]))
            pe = e if pe is None else max(pe, e)


if __name__ == "__main__":
    main()
