#!/usr/bin/env python3
"""Idle gaps and per-kernel busy time inside the LAST `win` ms of a rocprofv3 kernel-trace database (a serial prover such as
Marlin's: where does the device wait for the host, which kernels run alone?).
    python tools/rocpd_gaps.py x.db [win_ms=80] [gap_us=100]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 80.0
gap_us = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
cur = db.cursor()
syms = {r[0]: re.sub(r"\(.*", "", r[1]) for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
g = [c for c in cols if c.startswith("grid_size")]
rows = cur.execute(f"select kernel_id, start, end, {'*'.join(g) if g else '64'} from rocpd_kernel_dispatch order by start").fetchall()
def short(n):
    n = re.sub(r"^_ZN3zkp\d*", "", n)
    n = re.sub(r"^\d*cfg_c\d+\d*", "", n)
    m = re.search(r"([a-z_0-9]+_kernel)", n)
    b = m.group(1) if m else n[:32]
    return b + (":G2" if "Fp2" in n else "")
end = max(r[2] for r in rows)
a = end - win * 1e6
sel = [(short(syms.get(k, str(k))), max(s, a), e, gsz) for k, s, e, gsz in rows if e > a]
ev = sorted([(s, 1, n) for n, s, e, _ in sel] + [(e, -1, n) for n, s, e, _ in sel])
busy, depth, last, gaps, solo = 0, 0, a, [], {}
running = {}
prev_end_name = "-"
for t, d, n in ev:
    if depth > 0:
        busy += t - last
        if depth == 1:
            k = next(iter([x for x, c in running.items() if c > 0]), "?")
            solo[k] = solo.get(k, 0) + (t - last)
    elif t - last > gap_us * 1e3:
        gaps.append((last, t, prev_end_name, n))
    depth += d
    running[n] = running.get(n, 0) + d
    if d < 0:
        prev_end_name = n
    last = t
print(f"window {win:.1f} ms: union coverage {busy / (win * 1e6):.3f}; idle {win - busy / 1e6:.2f} ms")
print(f"gaps > {gap_us:.0f} us: {len(gaps)}, total {sum(e - s for s, e, _, _ in gaps) / 1e6:.2f} ms")
for s, e, p, n in gaps:
    print(f"   t={(s - a) / 1e6:8.3f} ms  idle {(e - s) / 1e3:8.1f} us   after {p:36s} before {n}")
tot = {}
for n, s, e, _ in sel:
    c = tot.setdefault(n, [0, 0.0])
    c[0] += 1
    c[1] += e - s
print("busy time by kernel (sum of durations) and time running ALONE:")
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"   {n:44s} x{c:5d}  {t / 1e6:8.3f} ms   alone {solo.get(n, 0) / 1e6:8.3f} ms")
