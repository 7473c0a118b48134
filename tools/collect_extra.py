#!/usr/bin/env python3
"""Second half of the round's profile collection (after tools/collect_profiles.py): request-size PMC table, timeline overlap +
contention table, the driver-flag and table-budget bench lines.   python tools/collect_extra.py r03"""
import json, re, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r03"
P, G = "profiles/", "gpurun_out/"


def last(p):
    return open(p).read().strip().split("\n")[-1]


body = open(G + f"prof_{R}_bytes/counts.txt").read()
m = re.search(r"cfg_c0117accumulate_kernel\S*\s+(\d+)\s+([\d.]+)\s+(\d+)", body)
ms = re.search(r"sort_scatter_staged\S*\s+(\d+)\s+([\d.]+)(?:\s+\d+){5}\s+(\d+)\s+(\d+)\s+(\d+)", body)
hdr = ["# rocprofv3 --kernel-trace --pmc <counters> (five separate passes, tools/pmc_bytes.sh) on",
       "# python bench.py --steps 3 --warmup 1 --no-pipeline --no-cpu-baseline --no-marlin   (MI355X, 2^20 BN254, final build of the round)",
       "# L2 (TCC) memory-side requests by size, per launch (rows are per counter instance: x32 for the device total is already",
       "# applied by the _sum counters).  Bytes = 32*RDREQ_32B + 64*RDREQ_64B + 128*RDREQ_128B ; writes 64*WRREQ_64B + 32*(WRREQ - WRREQ_64B).",
       f"# Round 2 -> round 3, per launch:  G1 accumulate reads 19.58 M -> {int(m.group(3)) / 1e6:.2f} M requests (all 128 B; this run's launch mix),",
       "#   sort_scatter writes 11.31 M (9.05 M 32-B + 2.26 M 64-B = 434 MB for ~110-125 MB of entries) -> see sort_scatter_staged below (~4.2 M, ~200 MB)."]
open(P + f"{R}_pmc_request_sizes.txt", "w").write("\n".join(hdr) + "\n" + body)


def table(path, avg_col):
    d = {}
    for l in open(path):
        f = l.split()
        if len(f) < 4 or not re.match(r"\d+$", f[1]):
            continue
        d[f[0]] = (int(f[1]), float(f[avg_col]))
    return d


iso = table(G + f"prof_{R}_bytes/counts.txt", 2)
pip = table(G + f"prof_{R}_overlap/kernel_stats.txt", 3)
rows = [("sort_hist", "sort_hist_kernel"), ("sort_scatter (staged)", "sort_scatter_staged"), ("sort_bin", "sort_bin_kernel"),
        ("ntt_pass2", "ntt_pass2_kernel"), ("segsum<G1>", "segsum_kernelINS_2Fp"), ("segsum<G2>", "segsum_kernelINS_3Fp2"),
        ("pair<G1>", "pair_kernelINS_2Fp"), ("pair<G2>", "pair_kernelINS_3Fp2"), ("accumulate G1", "cfg_c0117accumulate_kernel"),
        ("accumulate G2", "cfg_c0217accumulate_kernel")]


def find(d, key):
    for k, v in d.items():
        if key in k:
            return v
    return None


traced = json.loads(last(G + f"prof_{R}_overlap/trace_bench.json"))["value"]
full = json.loads(last(G + "bench_full.json"))["value"]
out = ["# rocprofv3 --kernel-trace -- python bench.py --no-cpu-baseline --no-marlin --steps 48 --warmup 8 (tools/trace_overlap.sh; "
       "tools/rocpd_overlap.py, tools/rocpd_fill.py); final build of the round",
       open(G + f"prof_{R}_overlap/overlap.txt").read().rstrip(), open(G + f"prof_{R}_overlap/fill.txt").read().rstrip(), "",
       "# contention: average kernel duration inside the pipelined run (8 proofs in flight, kernel trace: tracing itself slows the run,",
       f"# {traced:.1f} instead of {full:.1f} proofs/s) vs isolated (PMC passes of a --no-pipeline run)",
       "# kernel                 | isolated us | pipelined us | ratio      (round 2: sort_bin 79 -> 380, sort_hist 68 -> 296, ntt_pass 58 -> 130, "
       "segsum<G1> 121 -> 292, pair<G2> 63 -> 222)"]
for name, key in rows:
    a, b = find(iso, key), find(pip, key)
    out.append(f"{name:24s} {a[1]:10.1f} {b[1]:11.1f} {b[1] / a[1]:8.2f}" if a and b else f"{name:24s} missing")
open(P + f"{R}_overlap_contention.txt", "w").write("\n".join(out) + "\n")
open(P + f"{R}_bench_bn254_2p20_driver_flags.json", "w").write(last(G + "bench_driver_flags.json") + "\n")
open(P + f"{R}_bench_bn254_2p24_table_budget_24GB.json", "w").write(last(G + "bn24_budget24.json") + "\n")
print("\n".join(out[-11:]))
