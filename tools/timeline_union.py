#!/usr/bin/env python3
"""How much of a step's wall time the GPU runs at least one of our kernels (streams overlap, so busy time is the UNION of the kernel
intervals), from a rocprofv3 --kernel-trace rocpd database.  Steps are split at the temporal kernels.
    rocprofv3 --kernel-trace -d out -o t -- tools/shape_bench.sh 848x480;  python tools/timeline_union.py out/**/t_results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels where name like '%cvvdp::%' order by start").fetchall()
steps, cur = [], []
for n, s, e in rows:
    if "k_fir" in n and cur:
        steps.append(cur)
        cur = []
    cur.append((n, s, e))
steps.append(cur)
tail = steps[-6:-1]
for st in tail:
    span = max(e for _, _, e in st) - st[0][1]
    iv = sorted((s, e) for _, s, e in st)
    union, hi = 0, iv[0][0]
    gaps = []
    for s, e in iv:
        if s > hi:
            gaps.append((s - hi) / 1e3)
            hi = s
        if e > hi:
            union += e - hi
            hi = e
    print("step: %d kernels, span %.3f ms, GPU busy (union) %.3f ms = %.0f %%, %d gaps, largest %s us" %
          (len(st), span / 1e6, union / 1e6, 100.0 * union / span, len(gaps), [round(g, 1) for g in sorted(gaps)[-5:]]))
if len(steps) > 2:
    a, b = steps[-3], steps[-2]
    print("step to step: %.3f ms between the starts of consecutive temporal kernels" % ((b[0][1] - a[0][1]) / 1e6))
