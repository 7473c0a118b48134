#!/usr/bin/env python3
"""Window-size sweep of ONE stand-alone resident-table G1 MSM (BASELINE.json's second metric half: MSM G1 Mop/s).

    python tools/msm_window_sweep.py [curve] [log_n ...]        -> one table per size: c, windows, ms per MSM, Mop/s, phase times

c = 20 was chosen for the pipelined Groth16 prover (reduction tails overlap other MSMs' accumulates).  A lone MSM pays its
bucket sort and its reduction tail serially, and both grow with the bucket count: the window size is a per-context
setting (zkp_ctx_config.msm_window_bits, msm.hip pick_window_bits), so every row creates a context and uploads its own window tables.  Scalars uniform, resident in HBM; the time is the host
wall of zkp_msm_g1_dev (result read back each call), best and median of `runs`."""
import os
import statistics
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ckb_zkp_amd import codec
from ckb_zkp_amd.api import Context
from ckb_zkp_amd.params import get_curve

curve = sys.argv[1] if len(sys.argv) > 1 else "bn254"
sizes = [int(x) for x in sys.argv[2:]] or [18, 20, 22]
runs = int(os.environ.get("SWEEP_RUNS", "20"))
ctx = Context(0)
c = get_curve(curve)
gen, _ = codec.g1_to_mont([c.g1], c)
first = None
for log_n in sizes:
    n = (1 << log_n) - 1
    rng = np.random.default_rng(log_n)
    d = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    d[:, 3] >>= np.uint64(4)
    xy, inf = ctx.fixed_base_mul(c, 1, gen, d)
    k = np.frombuffer(rng.bytes(32 * n), dtype=np.uint64).reshape(n, 4).copy()
    k[:, 3] = rng.integers(0, c.r >> 192, size=n, dtype=np.uint64)
    k_dev = ctx.to_device(k)
    print(f"# {curve} G1, n = 2^{log_n} - 1, {runs} runs per row, uniform scalars resident in HBM")
    print(f"{'ZKP_MSM_C':>10} {'best ms':>9} {'median ms':>10} {'Mop/s(best)':>12} {'Mop/s(med)':>11}  same result")
    ref = None
    rows = ["default"] + [str(x) for x in range(max(12, log_n - 6), min(22, log_n + int(os.environ.get("SWEEP_ABOVE", "1"))) + 1)]
    for cw in rows[:1] if os.environ.get("SWEEP_ONLY_DEFAULT") else rows:
        # the window size is a per-context setting since ABI 0.6 (zkp_ctx_config.msm_window_bits): one context per row
        row_ctx = ctx if cw == "default" else Context(0, dict(msm_window_bits=int(cw)))
        try:
            bases = row_ctx.upload_bases(c, 1, xy, inf)
        except Exception as e:                      # a plan the library refuses (n * W too large, ...)
            print(f"{cw:>10}  upload failed: {e!r}")
            continue
        for _ in range(3):
            out = bases.msm_dev(k_dev, n)
        ts = []
        for _ in range(runs):
            t = time.perf_counter()
            out = bases.msm_dev(k_dev, n)
            ts.append((time.perf_counter() - t) * 1e3)
        aff = ctx.into_affine(c, 1, out)
        key = (aff[0].tobytes(), bool(aff[1]))
        ref = ref or key
        b, m = min(ts), statistics.median(ts)
        print(f"{cw:>10} {b:9.3f} {m:10.3f} {n / b / 1e3:12.1f} {n / m / 1e3:11.1f}  {key == ref}")
        bases.free()
        if row_ctx is not ctx:
            row_ctx.close()
    ctx.dev_free(k_dev)
    print()
