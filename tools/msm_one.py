#!/usr/bin/env python3
"""one resident-table MSM (for rocprofv3 --kernel-trace --stats): msm_one.py curve group log_n [runs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ckb_zkp_amd import codec
from ckb_zkp_amd.api import Context
from ckb_zkp_amd.params import get_curve
curve, group, log_n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
runs = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ctx = Context(0)
c = get_curve(curve)
n = 1 << log_n
rng = np.random.default_rng(1)
d = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
d[:, 3] >>= np.uint64(4)
gen, _ = codec.g1_to_mont([c.g1], c) if group == 1 else codec.g2_to_mont([c.g2], c)
xy, inf = ctx.fixed_base_mul(c, group, gen, d)
k = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
k[:, 3] >>= np.uint64(4)
bases = ctx.upload_bases(c, group, xy, inf)
for i in range(runs):
    t = time.time()
    out = bases.msm(k)
    print("msm ms", (time.time() - t) * 1e3)
