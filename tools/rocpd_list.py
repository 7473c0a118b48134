#!/usr/bin/env python3
"""Every kernel dispatch inside the LAST `win` ms of a rocprofv3 kernel-trace database, in start order: offset, duration, queue /
stream, grid, name — the timeline of ONE blocking proof (tools/trace_latency.sh).
    python tools/rocpd_list.py x.db [win_ms=9.5] [min_us=0]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 9.5
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
cur = db.cursor()
syms = {r[0]: re.sub(r"\(.*", "", r[1]) for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
g = [c for c in cols if c.startswith("grid_size")]
w = [c for c in cols if c.startswith("workgroup_size")]
q = "queue_id" if "queue_id" in cols else "0"
st = "stream_id" if "stream_id" in cols else "0"
rows = cur.execute(f"select kernel_id, start, end, {'*'.join(g) if g else '0'}, {'*'.join(w) if w else '1'}, {q}, {st} from rocpd_kernel_dispatch order by start").fetchall()
def short(n):
    n = re.sub(r"^_ZN3zkp\d*", "", n)
    n = re.sub(r"^\d*cfg_c\d+\d*", "", n)
    m = re.search(r"([a-z_0-9]+_kernel)", n)
    b = m.group(1).replace("_kernel", "") if m else n[:28]
    return b + (":G2" if "Fp2" in n or "c02" in n or "c12" in n else "")
end = max(r[2] for r in rows)
a = end - win * 1e6
qs = {}
print(f"# last {win} ms; columns: start_ms dur_us stream grid_threads/wg name")
for k, s, e, gs, ws, qid, sid in rows:
    if e <= a or (e - s) / 1e3 < min_us:
        continue
    key = sid if sid else qid
    qi = qs.setdefault(key, len(qs))
    print(f"{(s - a) / 1e6:8.3f} {(e - s) / 1e3:9.1f}  s{qi:<2d} {gs // max(ws, 1):7d}x{ws:<4d} {short(syms.get(k, str(k)))}")
