#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace [--stats] [--pmc ...]) as text:
per-kernel launch count, total / average / min / max duration and share of GPU time, and (if
present) per-kernel PMC counter sums.  Usage: rocpd_summary.py results.db [> profiles/xyz.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), max(lds_size), "
                       "min(grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z)), max(grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z)) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("# rocprofv3 kernel trace summary of %s" % path.split("/")[-1])
    print("%-92s %7s %12s %11s %11s %11s %6s %5s %5s %7s %s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds", "workgroups(min..max)"))
    for n, c, tot, avg, mn, mx, vg, sg, lds, g0, g1 in rows:
        nm = n if len(n) <= 92 else n[:89] + "..."
        print("%-92s %7d %12.3f %11.1f %11.1f %11.1f %6.2f %5s %5s %7s %s..%s" % (nm, c, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, vg, sg, lds, g0, g1))
    print("# total GPU kernel time: %.3f ms" % (total / 1e6))
    # the same kernel runs once per pyramid level: break the cvvdp kernels down by launch size (level 0 = most workgroups)
    rows2 = cur.execute("select name, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z) as wg, count(*), avg(duration), min(duration), max(duration) "
                        "from kernels where name like '%cvvdp::%' group by name, wg having avg(duration) > 100000 order by avg(duration) desc").fetchall()
    if rows2:
        print("\n# cvvdp kernels by launch size (launches averaging > 0.1 ms)")
        print("%-72s %10s %7s %11s %11s %11s" % ("kernel", "workgroups", "calls", "avg_us", "min_us", "max_us"))
        for n, wg, c, avg, mn, mx in rows2:
            nm = n if len(n) <= 72 else n[:69] + "..."
            print("%-72s %10d %7d %11.1f %11.1f %11.1f" % (nm, wg, c, avg / 1e3, mn / 1e3, mx / 1e3))
    try:
        pm = cur.execute("select p.name, k.grid_x*k.grid_y*k.grid_z/(k.workgroup_x*k.workgroup_y*k.workgroup_z) as wg, p.counter_name, "
                         "count(distinct p.dispatch_id), sum(p.counter_value) from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id "
                         "group by p.name, wg, p.counter_name order by sum(p.counter_value) desc").fetchall()
    except sqlite3.Error as e:
        pm = []
    if pm:
        print("\n# PMC counters (sum over dispatches; per-dispatch = sum / dispatches)")
        print("%-72s %10s %-20s %9s %18s %18s" % ("kernel", "workgroups", "counter", "dispatch", "sum", "per_dispatch"))
        for n, wg, cn, c, v in pm:
            nm = n if len(n) <= 72 else n[:69] + "..."
            print("%-72s %10d %-20s %9d %18.1f %18.1f" % (nm, wg, cn, c, v, v / c))


if __name__ == "__main__":
    main(sys.argv[1])
