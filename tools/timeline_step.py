"""Start time and duration of every kernel of the last complete step of a rocprofv3 --kernel-trace run of bench.py (steps are split at
the temporal kernels):   python tools/timeline_step.py <dir>/**/t_results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels where name like '%cvvdp::%' order by start").fetchall()
steps, cur = [], []
for n, s, e in rows:
    if "k_fir" in n and cur:
        steps.append(cur); cur = []
    cur.append((n, s, e))
steps.append(cur)
st = steps[-2]
t0 = st[0][1]
for n, s, e in st:
    print("%-60s start %8.3f  dur %8.3f ms" % (n.replace("void cvvdp::","").split("(")[0][:60], (s - t0) / 1e6, (e - s) / 1e6))
