#!/usr/bin/env python3
"""Per-kernel PMC totals from a rocprofv3 rocpd database (one --pmc pass).  python tools/rocpd_pmc.py x.db [filter]
Values of FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3); MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts 128-B
requests at 64 B, i.e. reports 1/2 of the bytes of 16-B-per-lane reads -> the `x2` column applies that correction."""
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
print(f"{'kernel':80s} {'counter':12s} {'calls':>6s} {'sum_KiB':>14s} {'MiB/call':>10s} {'x2 MiB/call':>12s} {'avg_us':>9s}")
for (name, c), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:80]:80s} {c:12s} {a[0]:6d} {a[1]:14.1f} {a[1]/a[0]/1024:10.2f} {2*a[1]/a[0]/1024:12.2f} {a[2]/a[0]/1e3:9.1f}")
