#!/usr/bin/env python3
"""Kernel trace of rocprofv3 (--kernel-trace, *_kernel_trace.csv) -> how busy the GPU was: the union of the kernel intervals
against the span of the trace, the sum of the durations (overlap counted twice) and the per-kernel totals.  Used for the
in-process multi-rank measurements (profiles/inproc_weak.py), where several ranks' streams share one GPU.

    python profiles/tools/trace_union.py <kernel_trace.csv> [--skip-first-ms 0] [--top 14]
"""
import argparse
import csv
import re
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--top", type=int, default=14)
    ap.add_argument("--last-ms", type=float, default=0.0, help="only the last so many milliseconds of the trace (the timed region)")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "0"))))
    rows.sort()
    if a.last_ms > 0:
        t_end = max(r[1] for r in rows)
        rows = [r for r in rows if r[0] >= t_end - a.last_ms * 1e6]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    for s, e, _, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    total = sum(e - s for s, e, _, _ in rows)
    per = defaultdict(lambda: [0, 0])
    for s, e, k, _ in rows:
        k = re.sub(r"^void ", "", k)
        per[k][0] += e - s
        per[k][1] += 1
    print("span %.2f ms, GPU busy (union) %.2f ms = %.1f %%, sum of kernel durations %.2f ms (overlap factor %.2f), %d launches, %d queues"
          % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), total / 1e6, total / busy, len(rows), len({r[3] for r in rows})))
    for k, (t, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[: a.top]:
        print("%9.2f ms %6d x %8.1f us  %s" % (t / 1e6, n, t / n / 1e3, k[:110]))


if __name__ == "__main__":
    main()
