#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 kernel-trace database: inside the window [t0, t1] (ms from the first dispatch; default =
the busiest 500 ms) report, per kernel class, the busy time (sum of durations), and for the whole device the UNION coverage
(fraction of wall time with at least one kernel running) and the mean number of concurrently running kernels.
    python tools/rocpd_overlap.py x.db [t0_ms t1_ms]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
syms = {r[0]: re.sub(r"\(.*", "", r[1]) for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
rows = cur.execute("select kernel_id, start, end from rocpd_kernel_dispatch order by start").fetchall()
T0 = rows[0][1]
if len(sys.argv) > 3:
    a, b = T0 + float(sys.argv[2]) * 1e6, T0 + float(sys.argv[3]) * 1e6
else:
    # window: the 500 ms with the most accumulate_kernel launches
    acc = [s for k, s, e in rows if "accumulate_kernel" in syms.get(k, "")]
    best, j = (0, acc[0]), 0
    for i, s in enumerate(acc):
        while acc[j] < s - 5e8:
            j += 1
        if i - j > best[0]:
            best = (i - j, acc[j])
    a, b = best[1], best[1] + 5e8
sel = [(k, max(s, a), min(e, b)) for k, s, e in rows if e > a and s < b]
wall = b - a
cls = {}
for k, s, e in sel:
    n = syms.get(k, str(k))
    key = ("accumulate" if "accumulate_kernel" in n else "ntt" if "ntt_pass" in n else "sort" if "sort_" in n else
           "reduce(pair/segsum/final/combine)" if any(x in n for x in ("pair_kernel", "segsum", "final", "combine")) else
           "assemble" if "assemble" in n else "other")
    c = cls.setdefault(key, [0, 0])
    c[0] += 1
    c[1] += e - s
ev = sorted([(s, 1) for _, s, e in sel] + [(e, -1) for _, s, e in sel])
busy, area, depth, last = 0, 0, 0, a
for t, d in ev:
    if depth > 0:
        busy += t - last
    area += depth * (t - last)
    depth += d
    last = t
print(f"window {wall/1e6:.1f} ms: union coverage {busy/wall:.3f}, mean concurrency {area/wall:.2f}")
for k, (n, t) in sorted(cls.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:36s} launches {n:6d}  busy {t/1e6:9.2f} ms  = {t/wall:.3f} of the window")


def coverage(pred):
    iv = [(s, e) for k, s, e in sel if pred(syms.get(k, ""))]
    ev2 = sorted([(s, 1) for s, e in iv] + [(e, -1) for s, e in iv])
    busy2, area2, d2, last2 = 0, 0, 0, a
    for t, d in ev2:
        if d2 > 0:
            busy2 += t - last2
        area2 += d2 * (t - last2)
        d2 += d
        last2 = t
    return busy2 / wall, area2 / wall


# how much of the window has at least one VALU-bound (machine-filling) kernel in flight, and how many at once
for label, pred in (("accumulate", lambda n: "accumulate_kernel" in n),
                    ("accumulate|ntt_pass", lambda n: "accumulate_kernel" in n or "ntt_pass" in n),
                    ("accumulate|ntt|pair|segsum", lambda n: any(x in n for x in ("accumulate_kernel", "ntt_pass", "pair_kernel", "segsum")))):
    c, m = coverage(pred)
    print(f"  >= 1 {label:28s} in flight: {c:.3f} of the window, mean concurrency {m:.2f}")
