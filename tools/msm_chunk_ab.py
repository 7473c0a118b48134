#!/usr/bin/env python3
"""Stand-alone 2^k G1 MSM (resident bases + resident canonical scalars, zkp_msm_g1_dev) under chunk settings:
   python tools/msm_chunk_ab.py [log_n=20]      env: ZKP_MSM_CHUNK (points per chunk; MSMs >= 2 chunks are chunked), ZKP_MSM_CHUNK_FIRST"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ckb_zkp_amd import codec
from ckb_zkp_amd.api import Context
from ckb_zkp_amd.params import get_curve
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = Context(0)
c = get_curve("bn254")
n = (1 << log_n) - 1
rng = np.random.default_rng(1)
d = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
d[:, 3] >>= np.uint64(4)
gen, _ = codec.g1_to_mont([c.g1], c)
xy, inf = ctx.fixed_base_mul(c, 1, gen, d)
k = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
k[:, 3] >>= np.uint64(4)
bases = ctx.upload_bases(c, 1, xy, inf)
kd = ctx.to_device(k)
ref = None
for _ in range(3):
    out = bases.msm_dev(kd, n)
t = time.perf_counter()
R = 20
for _ in range(R):
    out = bases.msm_dev(kd, n)
dt = (time.perf_counter() - t) / R
xy_, inf_ = ctx.into_affine(c, 1, out)
print(f"chunk={os.environ.get('ZKP_MSM_CHUNK','default')} first={os.environ.get('ZKP_MSM_CHUNK_FIRST','default')}: {dt*1e3:.3f} ms = {n/dt/1e6:.1f} Mop/s  x0={int(xy_[0]):x}")
