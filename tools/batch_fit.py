#!/usr/bin/env python3
"""Fill / drain cost of the pipelined prover: wall time T(n) of ONE zkp_groth16_prove_batch call for several n, and the least-squares
line T = a + b n (a = what a batch pays for filling and draining the lanes, b = steady-state time per proof).
    python tools/batch_fit.py [log_n=20] [ns=8,12,16,20,24,32,48,64] [reps=3]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from ckb_zkp_amd import codec, groth16
from ckb_zkp_amd.api import Context
from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain
from ckb_zkp_amd.params import get_curve
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ns = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8,12,16,20,24,32,48,64").split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
c = get_curve("bn254")
ctx = Context(0)
inst = mimc_chain_instance(c, samples_for_domain(log_n))
params = groth16.generate_parameters(ctx, c, inst, alpha=11, beta=13, gamma=17, delta=19, tau=23)
pk = groth16.ProvingKey(ctx, params, inst)
z_dev = ctx.to_device(codec.fr_to_mont(inst.z, c).reshape(-1, 4))
rng = np.random.default_rng(7)
def rnd(k):
    return np.stack([codec.fr_to_mont([int.from_bytes(rng.bytes(32), "little") % c.r], c)[0] for _ in range(k)])
pk.prove_batch_raw([z_dev] * 16, rnd(16), rnd(16))
res = {}
for rep in range(reps):
    for n in ns:
        r, s = rnd(n), rnd(n)
        ctx.sync()
        t0 = time.perf_counter()
        pk.prove_batch_raw([z_dev] * n, r, s)
        ctx.sync()
        res.setdefault(n, []).append((time.perf_counter() - t0) * 1e3)
xs = np.array(ns, dtype=float)
ys = np.array([min(res[n]) for n in ns])
b, a = np.polyfit(xs, ys, 1)
print("n      :", ns)
print("T(n) ms:", [round(float(y), 2) for y in ys])
print(f"fit T = {a:.2f} + {b:.3f} n  ->  {1e3 / b:.1f} proofs/s steady, fill+drain {a:.2f} ms; at n=20: {20e3 / (a + 20 * b):.1f} proofs/s")
