#!/usr/bin/env python3
"""Round-5 profile collection (after `bash tools/profile_r5.sh` on the GPU box): tools/collect_profiles.py for the Groth16 part, then
  profiles/r05_pmc_marlin_accumulate.json   FETCH_SIZE x2 + WRITE_SIZE of the accumulate kernel inside the Marlin prover (what
                                            bench.py's marlin_config4.roofline.traffic quotes)
  profiles/r05_pmc_ntt.json                 VALU issue share of ntt_pass2_kernel (bench.py roofline_ntt.valu_busy_recorded)
  profiles/r05_marlin_trace.txt, r05_pmc_ntt_pass.txt, r05_marlin_native_config4.json, r05_bench_bn254_2p20_driver_flags.json
    python tools/collect_r5.py"""
import json, os, re, subprocess, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r05"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
P, G = root + "profiles/", root + "gpurun_out/"
subprocess.run([sys.executable, root + "tools/collect_profiles.py", R, G + f"prof_{R}", G.rstrip("/")], check=True)


def last(p):
    return open(p).read().strip().split("\n")[-1]


def full_record(path):
    """since round 6 a bench run prints a compact line and writes the full record next to it (ZKP_BENCH_DETAIL=<path minus .json>.detail.json,
    tools/profile_r6.sh): the full record when it exists, else the line"""
    d = path[:-5] + ".detail.json"
    return open(d).read().strip() if os.path.exists(d) else last(path)


def acc_row(path):
    for l in open(path):
        if "accumulate_kernel" in l and "redo" not in l:
            f = l.split()
            return int(f[2]), float(f[3]), float(f[-1])            # calls, sum KiB, avg us
    raise SystemExit("accumulate_kernel not in " + path)


M = G + f"prof_{R}_marlin/"
nf, kf, uf = acc_row(M + "pmc_FETCH_SIZE.txt")
nw, kw, _ = acc_row(M + "pmc_WRITE_SIZE.txt")
line = json.loads(full_record(M + "pmc_FETCH_SIZE.json"))
m = line.get("marlin", line)
proofs = int(m.get("runs", 1)) + 1                              # bench_marlin: one warm + `runs` timed proofs
idx_launches = 12                                              # zkp_marlin_index_commit: 12 MSMs, once per run
per_proof = (nf - idx_launches) / proofs
per_launch = (2 * kf / nf + kw / nw) * 1024
cp, op = m["counts"]["commit_points"], m["counts"]["open_points"]
json.dump({
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --workload marlin "
              "--no-cpu-baseline --steps 1` (tools/profile_" + R + ".sh), MI355X; rows in profiles/" + R + "_marlin_trace.txt",
    "kernel": "accumulate_kernel (G1) inside zkp_marlin_index_commit + zkp_marlin_prove",
    "calls_in_run": nf, "proofs_in_run": proofs, "accumulate_launches_per_proof": round(per_proof, 1),
    "fetch_kib_raw_per_launch": round(kf / nf, 1), "write_kib_per_launch": round(kw / nw, 1), "avg_launch_us": uf,
    "gfx950_fetch_correction": 2.0, "traffic_bytes_per_launch": int(per_launch),
    "traffic_bytes_per_proof_all_msms": int(per_launch * per_proof),
    "traffic_bytes_per_proof_commit_msms": int(per_launch * per_proof * cp / (cp + op)),
    "note": "per launch = FETCH_SIZE x 2 + WRITE_SIZE averaged over the accumulate launches of the run; per proof = x launches of one "
            "proof; commit share = x commit_points / (commit_points + open_points) (the opening MSMs run the same kernel)"},
    open(P + f"{R}_pmc_marlin_accumulate.json", "w"), indent=1)

# NTT: VALU issue share from the counter passes of tools/pmc_ntt.sh (counts.txt: per-kernel averages per counter)
cnt = open(G + f"prof_{R}_ntt20/counts.txt").read()
open(P + f"{R}_pmc_ntt_pass.txt", "w").write(
    "# rocprofv3 --kernel-trace --pmc <counters> (six separate passes, tools/pmc_ntt.sh) on python tools/ntt_time.py bn254 20, MI355X, final build\n" + cnt)


rows = [l.split() for l in cnt.splitlines() if l.strip()]
hdr = rows[0]
row = next(r_ for r_ in rows[1:] if r_[0].startswith("ntt_pass2"))
cols = dict(zip(hdr[3:], row[3:]))                             # column names are right-truncated to 16 characters


def counter(name):
    for k, v in cols.items():
        if name.endswith(k) or k.endswith(name[-len(k):]):
            return float(v)
    return None


vals = {k: counter(k) for k in ("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")}
vals["avg_us"] = float(row[2])
json.dump({"source": f"profiles/{R}_pmc_ntt_pass.txt (tools/pmc_ntt.sh, 2^20, bn254; per counter instance, averaged over the launches)",
           "counters_avg_per_launch": vals,
           # a counter instance = 8 CUs = 32 SIMDs; a VALU wave-instruction occupies its SIMD's issue port for 4 cycles
           "valu_issue_share": (round(vals["SQ_INSTS_VALU"] * 4 / 32 / vals["SQ_BUSY_CYCLES"], 3)
                                if vals.get("SQ_INSTS_VALU") and vals.get("SQ_BUSY_CYCLES") else None),
           "valu_lane_instructions_per_element_per_pass": (round(vals["SQ_INSTS_VALU"] * 64 * 32 / (1 << 20), 1)
                                                           if vals.get("SQ_INSTS_VALU") else None)},
          open(P + f"{R}_pmc_ntt.json", "w"), indent=1)

tr = [f"# rocprofv3 --kernel-trace of python bench.py --workload marlin --no-cpu-baseline (tools/trace_marlin.sh, WIN=66): last proof of the run, MI355X, build of round {R}",
      open(M + "gaps.txt").read().rstrip(), "",
      "# accumulate kernel inside the Marlin proof, separate PMC passes (x2 = gfx950 FETCH_SIZE correction)",
      open(M + "pmc_FETCH_SIZE.txt").read().rstrip(), open(M + "pmc_WRITE_SIZE.txt").read().rstrip(), "",
      "# 1 ms timeline (top 4 kernels by busy time per bucket)", open(M + "timeline.txt").read().rstrip()]
open(P + f"{R}_marlin_trace.txt", "w").write("\n".join(tr) + "\n")
for a, b in (("marlin.json", f"{R}_marlin_native_config4.json"), ("bench_driver_flags.json", f"{R}_bench_bn254_2p20_driver_flags.json")):
    if os.path.exists(G + a):
        open(P + b, "w").write(full_record(G + a) + "\n")
print(open(P + f"{R}_pmc_marlin_accumulate.json").read())
print(open(P + f"{R}_pmc_ntt.json").read())
