#!/usr/bin/env python
"""Turn rocprofv3's rocpd SQLite output (gpurun_out/<dir>/*_results.db) into small CSV summaries
that can be committed under profiles/.

  python profiles/summarize_rocpd.py stats  gpurun_out/prof_stats/bench_results.db  profiles/r1_kernel_stats.csv
  python profiles/summarize_rocpd.py pmc    gpurun_out/prof_fetch/bench_results.db  profiles/r1_pmc_fetch.csv
"""
import csv
import sqlite3
import sys


def short(name):
    name = name.replace("void nepmi::nepmi_kernel", "nepmi_kernel").replace("nepmi::", "")
    cut = name.find(">(")
    return name[:cut + 1] if cut > 0 else name.split("(")[0]


def stats(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(scratch_size) "
        "from kernels group by name order by sum(end-start) desc").fetchall() if has(cur, "kernels") else []
    total = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent", "vgpr", "agpr", "sgpr", "scratch"])
        for r in rows:
            w.writerow([short(r[0]), r[1], "%.1f" % (r[2] / 1e3), "%.2f" % (r[3] / 1e3), "%.2f" % (r[4] / 1e3),
                        "%.2f" % (r[5] / 1e3), "%.2f" % (100.0 * r[2] / total), r[6], r[7], r[8], r[9]])


def has(cur, name):
    return bool(cur.execute("select 1 from sqlite_master where name=?", (name,)).fetchall())


def pmc(db, out):
    cur = sqlite3.connect(db).cursor()
    # one row per (dispatch, counter, dimension): sum the dimensions of a dispatch, then average
    rows = cur.execute(
        "select kernel_name, counter_name, count(distinct dispatch_id), sum(value), avg(duration) "
        "from counters_collection group by kernel_name, counter_name order by kernel_name").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "sum_per_dispatch", "avg_duration_us"])
        for r in rows:
            w.writerow([short(r[0]), r[1], r[2], "%.4f" % (r[3] / r[2]), "%.2f" % (r[4] / 1e3)])


def timeline(db, out):
    """Gaps on the device timeline: for every pair (kernel A ends, kernel B starts next on the queue) the mean idle time
    between them, over dispatches longer than 20 us (the discarded speculative launches of a run loop return at once)."""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    rows = [(short(n), s, e) for n, s, e in rows if e - s > 20000]
    gaps = {}
    busy = idle = 0
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        g = s1 - e0
        if g > 2000000:  # a host-side pause (rebuild, set-up), not a launch gap
            continue
        k = (n0, n1)
        a = gaps.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += g
        a[2] += e1 - s1
        busy += e1 - s1
        idle += max(g, 0)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel_before", "kernel_after", "count", "mean_gap_us", "mean_duration_after_us"])
        for (n0, n1), (c, g, d) in sorted(gaps.items(), key=lambda kv: -kv[1][0]):
            if c >= 5:
                w.writerow([n0, n1, c, "%.2f" % (g / c / 1e3), "%.2f" % (d / c / 1e3)])
        w.writerow(["TOTAL busy_us / idle_us / idle fraction", "", "", "%.1f" % (busy / 1e3), "%.1f / %.4f" % (idle / 1e3, idle / max(busy + idle, 1))])


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "timeline": timeline}[sys.argv[1]](sys.argv[2], sys.argv[3])
