#!/usr/bin/env python3
"""Per-kernel raw counter values (per call) from one or more rocprofv3 --pmc databases.
    python tools/rocpd_counts.py a.db [b.db ...] [--filter substr]"""
import re, sqlite3, sys
args = [a for a in sys.argv[1:] if not a.startswith("--")]
flt = ""
if "--filter" in sys.argv:
    flt = sys.argv[sys.argv.index("--filter") + 1]
    args = [a for a in args if a != flt]
tab = {}
for path in args:
    db = sqlite3.connect(path)
    cur = db.cursor()
    syms = {r[0]: r[1] for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
    pmc = {r[0]: r[1] for r in cur.execute("select id, name from rocpd_info_pmc")}
    rows = cur.execute("""select d.kernel_id, e.pmc_id, e.value, d.end - d.start from rocpd_pmc_event e
                          join rocpd_kernel_dispatch d on d.event_id = e.event_id""").fetchall()
    for kid, pid, val, dur in rows:
        name = re.sub(r"\(.*", "", syms.get(kid, str(kid)))
        name = re.sub(r"^_ZN3zkp\d*", "", name)[:44]
        if flt and flt not in name:
            continue
        a = tab.setdefault(name, {}).setdefault(pmc[pid], [0, 0.0, 0])
        a[0] += 1
        a[1] += val
        a[2] += dur
ctrs = sorted({c for v in tab.values() for c in v})
print(f"{'kernel':44s} {'calls':>6s} {'avg_us':>8s} " + " ".join(f"{c[-16:]:>16s}" for c in ctrs))
for name, v in sorted(tab.items(), key=lambda kv: -max(x[2] for x in kv[1].values())):
    any_c = next(iter(v.values()))
    print(f"{name:44s} {any_c[0]:6d} {any_c[2]/any_c[0]/1e3:8.1f} " + " ".join(f"{(v[c][1]/v[c][0]) if c in v else float('nan'):16.0f}" for c in ctrs))
