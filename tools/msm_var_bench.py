#!/usr/bin/env python3
"""Cold (variable-base, zkp_msm_g1_var) vs resident (window tables) G1 MSM at 2^k points: wall time per call incl. the
H2D of points + scalars for the cold path."""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ckb_zkp_amd import codec
from ckb_zkp_amd.api import Context
from ckb_zkp_amd.params import get_curve

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
group = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c = get_curve("bn254")
ctx = Context(0)
n = 1 << k
rng = np.random.default_rng(1)
d = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
d[:, 3] >>= np.uint64(4)
g_xy, _ = (codec.g1_to_mont([c.g1], c) if group == 1 else codec.g2_to_mont([c.g2], c))
xy, inf = ctx.fixed_base_mul(c, group, g_xy, d)
sc = np.frombuffer(rng.bytes(32 * n), dtype=np.uint64).reshape(-1, 4).copy()
sc[:, 3] &= np.uint64((1 << (c.r.bit_length() - 193)) - 1)
res = {}
for _ in range(2):
    out_v = ctx.msm_var(c, group, xy, None, sc)
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    out_v = ctx.msm_var(c, group, xy, None, sc)
res["var_ms"] = (time.perf_counter() - t0) / reps * 1e3
t0 = time.perf_counter()
b = ctx.upload_bases(c, group, xy, None)
res["upload_precompute_ms"] = (time.perf_counter() - t0) * 1e3
for _ in range(2):
    out_r = b.msm(sc)
t0 = time.perf_counter()
for _ in range(reps):
    out_r = b.msm(sc)
res["resident_host_scalars_ms"] = (time.perf_counter() - t0) / reps * 1e3
sd = ctx.to_device(sc)
b.msm_dev(sd, n)
t0 = time.perf_counter()
for _ in range(reps):
    b.msm_dev(sd, n)
res["resident_dev_scalars_ms"] = (time.perf_counter() - t0) / reps * 1e3
same = ctx.into_affine(c, group, out_v)[0].tolist() == ctx.into_affine(c, group, out_r)[0].tolist()
res.update(n=n, group=group, equal=same)
print(json.dumps(res))
