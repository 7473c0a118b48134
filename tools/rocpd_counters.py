#!/usr/bin/env python3
"""Per-kernel raw PMC sums from a rocprofv3 rocpd database: python tools/rocpd_counters.py x.db [kernel filter]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
syms = {r[0]: r[1] for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
pmc = {r[0]: r[1] for r in cur.execute("select id, name from rocpd_info_pmc")}
rows = cur.execute("""select d.kernel_id, e.pmc_id, e.value, d.end - d.start from rocpd_pmc_event e
                      join rocpd_kernel_dispatch d on d.event_id = e.event_id""").fetchall()
agg = {}
for kid, pid, val, dur in rows:
    name = re.sub(r"\(.*", "", syms.get(kid, str(kid)))
    if flt and flt not in name:
        continue
    a = agg.setdefault((name, pmc[pid]), [0, 0.0, 0])
    a[0] += 1
    a[1] += val
    a[2] += dur
print(f"{'kernel':70s} {'counter':24s} {'calls':>6s} {'sum':>18s} {'per_call':>16s} {'avg_us':>9s}")
for (name, c), a in sorted(agg.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    print(f"{name[:70]:70s} {c:24s} {a[0]:6d} {a[1]:18.1f} {a[1]/a[0]:16.1f} {a[2]/a[0]/1e3:9.1f}")
