#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table
(equivalent of `--stats` CSV): calls, total/avg/min/max duration.   python tools/rocpd_stats.py x.db [last_ms]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
syms = {r[0]: r[1] for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
rows = cur.execute("select kernel_id, start, end from rocpd_kernel_dispatch order by start").fetchall()
t_end = rows[-1][2]
last = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else None      # only dispatches in the last <ms>
agg = {}
for kid, s, e in rows:
    if last is not None and t_end - s > last:
        continue
    name = syms.get(kid, str(kid))
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    a = agg.setdefault(name, [0, 0, 1 << 62, 0])
    d = e - s
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:90]:90s} {a[0]:7d} {a[1]/1e6:10.3f} {a[1]/a[0]/1e3:10.1f} {a[2]/1e3:9.1f} {a[3]/1e3:9.1f} {100*a[1]/tot:6.2f}")
print(f"{'TOTAL':90s} {sum(a[0] for a in agg.values()):7d} {tot/1e6:10.3f}")
