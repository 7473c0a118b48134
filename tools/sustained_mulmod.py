#!/usr/bin/env python3
"""Is the multiplier microbenchmark (valu_roof's ceiling) a burst number?  Calls zkp_bench_mulmod back to back for ~4 s and prints the
rate of every call; run `rocm-smi --showclocks --showpower` next to it (tools/sustained_mulmod.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ckb_zkp_amd.api import Context
ctx = Context(0)
t0 = time.time()
out = []
while time.time() - t0 < float(sys.argv[1]) if len(sys.argv) > 1 else 4.0:
    out.append((round(time.time() - t0, 2), round(ctx.bench_mulmod("bn254", 1, True), 1)))
print("t_s, G products/s:", out[:3], "...", out[len(out) // 2 - 1:len(out) // 2 + 2], "...", out[-3:], "calls", len(out))
