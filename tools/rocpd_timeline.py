#!/usr/bin/env python3
"""Coarse timeline of the LAST `win` ms of a rocprofv3 kernel-trace database: per `step`-ms bucket the busy time of the kernels that
ran in it (top 4 by busy time) — where a serial prover (Marlin) spends its wall clock, phase by phase.
    python tools/rocpd_timeline.py x.db [win_ms=82] [step_ms=1.0]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 82.0
step = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
cur = db.cursor()
syms = {r[0]: re.sub(r"\(.*", "", r[1]) for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
rows = cur.execute("select kernel_id, start, end from rocpd_kernel_dispatch order by start").fetchall()
def short(n):
    n = re.sub(r"^_ZN3zkp\d*", "", n)
    n = re.sub(r"^\d*cfg_c\d+\d*", "", n)
    m = re.search(r"([a-z_0-9]+_kernel)", n)
    b = m.group(1).replace("_kernel", "") if m else n[:24]
    return b + (":G2" if "Fp2" in n else "")
end = max(r[2] for r in rows)
a = end - win * 1e6
nb = int(win / step + 0.999)
buckets = [dict() for _ in range(nb)]
for k, s, e in rows:
    if e <= a:
        continue
    s = max(s, a)
    n = short(syms.get(k, str(k)))
    b0, b1 = int((s - a) / (step * 1e6)), int((e - a - 1) / (step * 1e6))
    for b in range(b0, min(b1, nb - 1) + 1):
        lo, hi = a + b * step * 1e6, a + (b + 1) * step * 1e6
        buckets[b][n] = buckets[b].get(n, 0.0) + (min(e, hi) - max(s, lo)) / 1e6
for b, d in enumerate(buckets):
    top = sorted(d.items(), key=lambda kv: -kv[1])[:4]
    print(f"{b * step:6.1f} ms  sum {sum(d.values()):5.2f}  " + "  ".join(f"{n} {t:.2f}" for n, t in top))
