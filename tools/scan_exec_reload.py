#!/usr/bin/env python3
"""Scans hipcc --save-temps device assembly for the code shape behind the round-2 combine_kernel fault: a register reload
(v_accvgpr_read / scratch_load) sitting immediately BEFORE an `s_or_b64 exec, exec, ...` restore, i.e. executed under a
narrowed exec mask although the value is consumed after the mask is widened again.  Heuristic; prints candidates to inspect.
usage: scan_exec_reload.py file.s ..."""
import re, sys
for path in sys.argv[1:]:
    fn = None
    lines = open(path).read().split("\n")
    n = len(lines)
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            fn = m.group(1)
        if re.match(r"\s+s_or_b64 exec, exec,", l):
            # walk back over labels / comments / reloads
            j = i - 1
            reloads = []
            while j >= 0:
                t = lines[j].strip()
                if not t or t.startswith(";") or t.startswith(".LBB") or t.startswith("s_nop") or t.startswith("s_waitcnt"):
                    j -= 1
                    continue
                if t.startswith("v_accvgpr_read") or t.startswith("scratch_load"):
                    reloads.append((j + 1, t))
                    j -= 1
                    continue
                break
            if reloads:
                print(f"{path}:{i+1} {fn[:70] if fn else '?'}: {len(reloads)} reload(s) before exec restore, e.g. line {reloads[-1][0]}: {reloads[-1][1]}")
