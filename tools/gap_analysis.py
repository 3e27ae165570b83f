#!/usr/bin/env python3
"""Idle time between consecutive kernels of one bench step, from a rocprofv3 --kernel-trace rocpd database.
Usage: gap_analysis.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels where name like '%cvvdp::%' order by start").fetchall()
# split into steps at every FIR kernel
steps, cur = [], []
for n, s, e in rows:
    if "k_fir" in n and cur:
        steps.append(cur); cur = []
    cur.append((n, s, e))
steps.append(cur)
for st in steps[-2:]:
    busy = sum(e - s for _, s, e in st)
    span = st[-1][2] - st[0][1]
    print("step: %d kernels, span %.3f ms, busy %.3f ms, idle %.3f ms" % (len(st), span / 1e6, busy / 1e6, (span - busy) / 1e6))
    prev = None
    for n, s, e in st:
        gap = (s - prev) / 1e3 if prev else 0.0
        print("   %-60s %9.1f us   gap before %7.1f us" % (n.split("(")[0][-60:], (e - s) / 1e3, gap))
        prev = e
