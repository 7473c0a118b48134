#!/usr/bin/env python3
"""Where the VALU instructions of a proof go: from the counts of tools/pmc_valu.sh (SQ_INSTS_VALU per launch and counter instance
of a --no-pipeline run) x launches per proof, converted to "ms of a fully busy vector ALU" with the G1 accumulate kernel's own rate
(instructions per ms of its isolated launches: that kernel is VALU-bound).   python tools/valu_budget.py gpurun_out/valu_r03b/counts.txt"""
import re, sys
rows = {}
hdr = None
for l in open(sys.argv[1]):
    f = l.split()
    if not hdr:
        hdr = f
        continue
    if len(f) < 9 or not re.match(r"\d+$", f[1]):
        continue
    rows[f[0]] = dict(calls=int(f[1]), us=float(f[2]), valu=float(f[6]))
def get(key):
    for k, v in rows.items():
        if key in k:
            return v
    raise SystemExit("missing " + key)
acc1 = get("cfg_c0117accumulate_kernel")
rate = acc1["valu"] / (acc1["us"] * 1e-3)          # instructions per ms (per counter instance) of a VALU-bound kernel
# launches per proof (prove path with bucket chaining and the shared level-1 pass)
plan = [("G1 accumulate (A, B1, L, H)", "cfg_c0117accumulate_kernel", 4), ("G2 accumulate (B2)", "cfg_c0217accumulate_kernel", 1),
        ("NTT passes (4 transforms x 3, evaluation-form key)", "ntt_pass2_kernel", 12),
        ("G1 segmented sums (3 reductions x 2)", "segsum_kernelINS_2Fp", 6), ("G1 pyramid pairs (3 x 8)", "pair_kernelINS_2Fp", 24),
        ("G2 segmented sums", "segsum_kernelINS_3Fp2", 2), ("G2 pyramid pairs", "pair_kernelINS_3Fp2", 8),
        ("level-1 scatter (A+B2+L group, H)", "sort_scatter_staged", 2), ("level-1 histogram", "sort_hist_kernel", 2),
        ("level-2 sort (A, B2, L, H)", "sort_bin_kernel", 4), ("sparse A z, B z", "csr_eval_kernel", 2)]
tot = 0.0
print(f"# calibration: G1 accumulate {acc1['valu']:.0f} VALU instructions / launch / counter instance in {acc1['us']:.1f} us -> {rate / 1e3:.1f} K per ms")
print(f"{'kernel':42s} {'launches':>8s} {'instr/launch':>13s} {'ms of VALU':>11s}")
for name, key, n in plan:
    r = get(key)
    ms = n * r["valu"] / rate
    tot += ms
    print(f"{name:42s} {n:8d} {r['valu']:13.0f} {ms:11.3f}")
print(f"{'total':42s} {'':8s} {'':13s} {tot:11.3f}   (single-lane tails, scans, task scheduling, assembly: < 0.1 ms more)")
