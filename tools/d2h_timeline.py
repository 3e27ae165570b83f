#!/usr/bin/env python3
"""Timeline of the heat-map D2H stream against the kernels (round 6) from a rocprofv3 rocpd database made with
    rocprofv3 --kernel-trace --memory-copy-trace -d DIR -o t -- python bench.py --workload 8k256pq --steps 2 --warmup 1 --cpu-frames 0 --no-profile --no-power-probe
Usage: tools/d2h_timeline.py results.db   -> large copies of the LAST step: start, duration, GB/s, idle gap of the copy engine before each;
kernel-busy time inside the same window; where the link sat idle."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
    mc = next((n for n in names if n == "memory_copies"), None) or next((n for n in names if "memory_cop" in n), None)
    if mc is None:
        print("no memory-copy view in this database; tables/views:", names)
        return
    cols = [r[1] for r in cur.execute(f"pragma table_info({mc})").fetchall()]
    st, en = ("start" if "start" in cols else "start_timestamp"), ("end" if "end" in cols else "end_timestamp")
    size = "size" if "size" in cols else next(c for c in cols if "size" in c or "bytes" in c)
    copies = cur.execute(f"select {st}, {en}, {size} from {mc} where {size} >= 67108864 order by {st}").fetchall()
    if not copies:
        print("no copies >= 64 MB; columns:", cols)
        for n in names:
            try:
                print("    %-40s %d rows" % (n, cur.execute(f"select count(*) from [{n}]").fetchone()[0]))
            except sqlite3.Error as e:
                print("    %-40s %s" % (n, e))
        for row in cur.execute(f"select name, count(*), min({size}), max({size}), sum({size}) from {mc} group by name").fetchall():
            print("   ", row)
        thr = cur.execute(f"select max({size}) from {mc}").fetchone()[0] or 0
        copies = cur.execute(f"select {st}, {en}, {size} from {mc} where {size} >= ? order by {st}", (thr // 2,)).fetchall()
        if not copies:
            return
    kcols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    ks, ke = ("start" if "start" in kcols else "start_timestamp"), ("end" if "end" in kcols else "end_timestamp")
    # the last step = the last run of copies separated from the ones before it by > 50 ms
    cut = 0
    for i in range(1, len(copies)):
        if copies[i][0] - copies[i - 1][1] > 50e6:
            cut = i
    last = copies[cut:]
    t0 = last[0][0]
    kern = cur.execute(f"select {ks}, {ke}, name from kernels where {ke} >= ? and {ks} <= ? order by {ks}", (t0 - 400e6, last[-1][1])).fetchall()
    # the step's first kernel: walk back from the first copy while kernels are closer than 5 ms to each other
    first_k = None
    for k in reversed([k for k in kern if k[0] < t0]):
        if first_k is None or first_k - k[1] < 5e6:
            first_k = k[0]
        else:
            break
    origin = first_k if first_k is not None else t0
    print(f"# {len(last)} copies >= 64 MB in the last step; time 0 = its first kernel; copy view '{mc}'")
    print("%4s %10s %10s %9s %8s %10s" % ("#", "start_ms", "dur_ms", "MB", "GB/s", "gap_ms"))
    prev_end, tot_b, tot_busy, tot_gap = None, 0, 0.0, 0.0
    for i, (a, b, n) in enumerate(last):
        gap = 0.0 if prev_end is None else max(0.0, (a - prev_end) / 1e6)
        print("%4d %10.2f %10.2f %9.0f %8.1f %10.2f" % (i, (a - origin) / 1e6, (b - a) / 1e6, n / 1e6, n / max(b - a, 1), gap))
        prev_end = b if prev_end is None else max(prev_end, b)
        tot_b += n; tot_busy += (b - a) / 1e6; tot_gap += gap
    span = (last[-1][1] - origin) / 1e6
    print(f"# first kernel -> last copy done: {span:.1f} ms; first copy starts at {(t0 - origin) / 1e6:.1f} ms; copy engine busy {tot_busy:.1f} ms "
          f"({tot_b / 1e9:.2f} GB at {tot_b / 1e6 / max(tot_busy, 1e-9):.1f} GB/s while copying), idle between copies {tot_gap:.1f} ms")
    # kernel-busy union inside [origin, last copy end]
    busy, cur_s, cur_e = 0.0, None, None
    for a, b, _ in sorted(k for k in kern if k[1] > origin):
        a = max(a, origin)
        if cur_e is None or a > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    if cur_e is not None:
        busy += cur_e - cur_s
    print(f"# kernels busy (union) in the same window: {busy / 1e6:.1f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
