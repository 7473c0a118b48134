#!/usr/bin/env python3
"""Copies the outputs of tools/profile_r.sh + the bench runs (gpurun_out/) into profiles/ with provenance headers and derives
profiles/<round>_pmc_accumulate.json (per-launch HBM traffic of the accumulate kernels).
    python tools/collect_profiles.py r02 gpurun_out/prof_r02e [gpurun_out]"""
import json, os, sys
rnd, src = sys.argv[1], sys.argv[2].rstrip("/") + "/"
out = sys.argv[3].rstrip("/") + "/" if len(sys.argv) > 3 else "gpurun_out/"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
P = root + "profiles/"


def last(path):
    return open(path).read().strip().split("\n")[-1]


def full_record(path):
    """since round 6 a bench run prints a compact line and writes the full record next to it (ZKP_BENCH_DETAIL=<path minus .json>.detail.json,
    tools/profile_r6.sh): the full record when it exists, else the line"""
    d = path[:-5] + ".detail.json"
    return open(d).read().strip() if os.path.exists(d) else last(path)


def row(path, kern):
    for l in open(path):
        if l.startswith(kern):
            f = l.split()
            return float(f[3]), int(f[2]), float(f[-1])
    raise SystemExit(f"{kern} not in {path}")


f, w = src + "pmc_FETCH_SIZE.txt", src + "pmc_WRITE_SIZE.txt"
A1, A2, NT = "_ZN3zkp7cfg_c0117accumulate_kernel", "_ZN3zkp7cfg_c0217accumulate_kernel", "_ZN3zkp16ntt_pass2_kernel"
fg1, n1, t1 = row(f, A1)
fg2, n2, t2 = row(f, A2)
wg1, _, _ = row(w, A1)
wg2, _, _ = row(w, A2)
fn, nn, _ = row(f, NT)
wn, _, _ = row(w, NT)
f1, f2, w1, w2 = fg1 / n1, fg2 / n2, wg1 / n1, wg2 / n2
traffic = int(((4 * (2 * f1 + w1) + (2 * f2 + w2)) / 5) * 1024)
json.dump({
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --steps 3 --warmup 1 --no-pipeline "
              "--no-cpu-baseline --no-marlin`, MI355X (tools/profile_r.sh, tools/collect_profiles.py); summaries in profiles/"
              + rnd + "_pmc_fetch_write_rocprofv3.txt",
    "kernel": "accumulate_kernel", "launches_per_proof": 5,
    "fetch_kib_raw_per_launch": {"g1": round(f1, 2), "g2": round(f2, 2)},
    "write_kib_per_launch": {"g1": round(w1, 2), "g2": round(w2, 2)},
    "avg_launch_us_in_pmc_run": {"g1": t1, "g2": t2},
    "gfx950_fetch_correction": 2.0,
    "calibration": f"ntt_pass2_kernel in the same run: FETCH_SIZE raw {fn / nn / 1024:.2f} MiB/launch (x2 = {2 * fn / nn / 1024:.1f} MiB) for a "
                   f"32 MiB vector + twiddle table minus cache hits, WRITE_SIZE {wn / nn / 1024:.2f} MiB/launch for a 32 MiB vector -> reads "
                   "need the x2 of MI355X_MICROARCH.md, writes are exact",
    "traffic_bytes_per_launch": traffic,
    "note": "average over the 5 accumulate launches of one proof (4 x G1 + 1 x G2): FETCH_SIZE x2 + WRITE_SIZE; the G1 figure averages "
            "the launch mix of the run (A / B1 / H / L queries and the standalone H-shaped MSMs); buckets / partial sums are written in "
            "the unsaturated layout (144 B per G1 point, 288 B per G2 point)"},
    open(P + rnd + "_pmc_accumulate.json", "w"), indent=1)
with open(P + rnd + "_pmc_fetch_write_rocprofv3.txt", "w") as o:
    o.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes; tools/profile_r.sh) on\n"
            "# python bench.py --steps 3 --warmup 1 --no-pipeline --no-cpu-baseline --no-marlin   (MI355X, 2^20 BN254, final build of the round)\n"
            "# x2 column = gfx950 correction for FETCH_SIZE (MI355X_MICROARCH.md); WRITE_SIZE is exact (use the MiB/call column)\n")
    o.write(open(f).read() + "\n" + open(w).read())
traced = json.loads(last(src + "stats_bench.json"))["value"]
full = json.loads(last(out + "bench_full.json"))["value"]
with open(P + rnd + "_kernel_stats_rocprofv3.txt", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-marlin   (MI355X, 2^20 BN254, final build of the\n"
            "# round; tools/profile_r.sh; 64+10 pipelined proofs on 8 lanes, 8+64 proofs with the witness in pinned host memory, 2 profiled\n"
            "# proofs, 28 standalone G1 MSMs, the NTT / multiplier-rate measurements of the JSON line; includes one-time key generation /\n"
            "# window-table precompute kernels.  Kernel tracing slows the pipelined run itself: the JSON line of THIS run says\n"
            f"# {traced:.1f} proofs/s, the untraced run {full:.1f} (profiles/{rnd}_bench_bn254_2p20.json))\n")
    o.write(open(src + "kernel_stats.txt").read())
for a, b in (("bench_full.json", "_bench_bn254_2p20.json"), ("bls22.json", "_bench_bls12_381_2p22.json"),
             ("bn24.json", "_bench_bn254_2p24_single_gpu.json"), ("marlin.json", "_marlin_native_config4.json")):
    if os.path.exists(out + a):
        open(P + rnd + b, "w").write(full_record(out + a) + "\n")
open(P + rnd + "_bench_under_kernel_trace.json", "w").write(full_record(src + "stats_bench.json") + "\n")
print("traffic_bytes_per_launch", traffic, "traced", traced, "untraced", full)
