#!/usr/bin/env python3
"""Blocking single proofs at 2^k (for a kernel-trace timeline of one proof): latency_one.py [curve=bn254] [log_n=20] [proofs=6]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ckb_zkp_amd import codec, groth16
from ckb_zkp_amd.api import Context
from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain
from ckb_zkp_amd.params import get_curve
c = get_curve(sys.argv[1] if len(sys.argv) > 1 else "bn254")
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ctx = Context(0)
inst = mimc_chain_instance(c, samples_for_domain(k))
params = groth16.generate_parameters(ctx, c, inst, alpha=11, beta=12, gamma=13, delta=14, tau=987654321)
pk = groth16.ProvingKey(ctx, params, inst)
z_dev = ctx.to_device(codec.fr_to_mont(inst.z, c).reshape(-1, 4))
rng = np.random.default_rng(5)
def fr():
    v = rng.integers(0, 1 << 61, size=4, dtype=np.uint64)
    return v
ts = []
for i in range(n):
    ctx.sync()
    time.sleep(float(os.environ.get("GAP_S", "0")))   # GAP_S=0.02: a visible idle gap in front of every proof (traces)
    t0 = time.perf_counter()
    pk.prove_raw(z_dev, fr(), fr(), z_on_device=True)
    ts.append((time.perf_counter() - t0) * 1e3)
print("latency ms", [round(t, 3) for t in ts])
