#!/usr/bin/env python3
"""Kernel timeline of the LAST `span_ms` of a rocprofv3 kernel-trace database, one line per kernel (start / end in us relative to
the window start, queue, name); kernels shorter than `min_us` are folded into a count per name.
    python tools/rocpd_timeline.py x.db [span_ms=12] [min_us=30]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
span = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
cur = db.cursor()
syms = {r[0]: re.sub(r"\(.*", "", r[1]) for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "queue_id" if "queue_id" in cols else None
rows = cur.execute(f"select kernel_id, start, end{', ' + qcol if qcol else ''} from rocpd_kernel_dispatch order by start").fetchall()
t1 = max(r[2] for r in rows)
t0 = t1 - span * 1e6
small = {}
for r in rows:
    if r[2] < t0:
        continue
    name = re.sub(r"^_ZN3zkp\d*", "", syms.get(r[0], "?"))[:60]
    dur = (r[2] - r[1]) / 1e3
    if dur < min_us:
        s = small.setdefault(name, [0, 0.0])
        s[0] += 1
        s[1] += dur
        continue
    print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - t0) / 1e3:9.1f} {dur:8.1f} q{r[3] if qcol else '?'}  {name}")
for n, (c, d) in sorted(small.items(), key=lambda x: -x[1][1]):
    print(f"   short: {c:4d} x {n}  total {d:.1f} us")
