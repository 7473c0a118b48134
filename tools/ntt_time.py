#!/usr/bin/env python3
"""Stand-alone NTT time on a resident vector (HIP events around 30 back-to-back transforms).
    python tools/ntt_time.py [curve=bn254] [log_ns=20,21,22,23]      (ZKP_ACCEL_LIB selects a build variant)"""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ckb_zkp_amd.api import Context, NTT_FFT, NTT_IFFT, NTT_COSET_FFT, NTT_COSET_IFFT
from ckb_zkp_amd.params import get_curve
c = get_curve(sys.argv[1] if len(sys.argv) > 1 else "bn254")
logs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "20,21,22,23").split(",")]
ctx = Context(0)
for k in logs:
    N = 1 << k
    buf = ctx.to_device(np.frombuffer(np.random.default_rng(3).bytes(32 * N), dtype=np.uint64).reshape(-1, 4) >> np.uint64(3))
    out = []
    for name, op in (("fft", NTT_FFT), ("ifft", NTT_IFFT), ("coset_fft", NTT_COSET_FFT), ("coset_ifft", NTT_COSET_IFFT)):
        for _ in range(3):
            ctx.ntt_dev(c, buf, k, op)
        ctx.timer_start()
        for _ in range(30):
            ctx.ntt_dev(c, buf, k, op)
        out.append(f"{name} {ctx.timer_stop_ms() / 30:.4f}")
    ctx.dev_free(buf)
    print(f"2^{k}: " + "  ".join(out) + " ms")
