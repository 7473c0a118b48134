#!/usr/bin/env python3
"""How full is the machine?  From a rocprofv3 kernel-trace database: every dispatch gets fill = min(1, waves / CAP) (waves from
its grid, CAP = 4096 = 256 CUs x 4 SIMDs x 4 waves, what a 100-130-VGPR kernel can hold); inside the busiest window the tool
integrates, over time, the SUM of fill over the kernels in flight and reports how much of the wall time is under-filled and which
kernels are in flight during that time.
    python tools/rocpd_fill.py x.db [window_ms=400]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
cur = db.cursor()
syms = {r[0]: re.sub(r"\(.*", "", r[1]) for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
g = [c for c in cols if c.startswith("grid_size")]
wg = [c for c in cols if c.startswith("workgroup_size")]
q = "queue_id" if "queue_id" in cols else "0"
rows = cur.execute(f"select kernel_id, start, end, {q}, {'*'.join(g) if g else '64'}, {'*'.join(wg) if wg else '64'} from rocpd_kernel_dispatch order by start").fetchall()
CAP = 4096.0
def short(n):
    n = re.sub(r"^_ZN3zkp\d*", "", n)
    n = re.sub(r"^7cfg_c\d+\d+", "", n)
    m = re.match(r"\d*([a-z_0-9]+?_kernel)", n)
    base = m.group(1) if m else n[:24]
    if "Fp2" in n or "cfg_c02" in syms.get(0, ""):
        base += ":G2" if "Fp2" in n else ""
    return base
acc = [r[1] for r in rows if "accumulate_kernel" in syms.get(r[0], "")]
best, j = (0, acc[0]), 0
for i, s in enumerate(acc):
    while acc[j] < s - win * 1e6:
        j += 1
    if i - j > best[0]:
        best = (i - j, acc[j])
a, b = best[1], best[1] + win * 1e6
ev = []
for k, s, e, qq, gs, ws in rows:
    if e <= a or s >= b:
        continue
    waves = max(1.0, gs / 64.0)
    ev.append((max(s, a), 1, k, min(1.0, waves / CAP)))
    ev.append((min(e, b), -1, k, min(1.0, waves / CAP)))
ev.sort(key=lambda x: (x[0], x[1]))
bins = [0.1, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 1e9]
tbin = [0.0] * len(bins)
under = {}
live = {}
fill, last = 0.0, a
for t, d, k, f in ev:
    dt = t - last
    if dt > 0:
        for i, x in enumerate(bins):
            if fill < x - 1e-9:
                tbin[i] += dt
                break
        if fill < 0.5 - 1e-9:
            for kk, c in live.items():
                if c > 0:
                    n = short(syms.get(kk, "?"))
                    under[n] = under.get(n, 0.0) + dt
            if not any(c > 0 for c in live.values()):
                under["(nothing)"] = under.get("(nothing)", 0.0) + dt
    fill += d * f
    live[k] = live.get(k, 0) + d
    last = t
wall = b - a
print(f"window {wall/1e6:.0f} ms, {best[0]} accumulate launches; sum of fill over kernels in flight (fill = min(1, waves / {CAP:.0f})):")
lo = 0.0
for x, t in zip(bins, tbin):
    print(f"   fill in [{lo:.2f}, {x if x < 1e8 else float('inf'):.2f}): {t / wall:.3f} of the time")
    lo = x
print("in flight while fill < 0.5 (share of the WINDOW):")
for n, t in sorted(under.items(), key=lambda kv: -kv[1])[:16]:
    print(f"   {n:40s} {t / wall:.3f}")
