#!/usr/bin/env python3
"""Wall time of consecutive zkp_groth16_prove_batch calls in one process (is the first timed batch after a short warm-up slower
than later ones?).   python tools/batch_seq.py [warmup=5] [n=20] [reps=8]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from ckb_zkp_amd import codec, groth16
from ckb_zkp_amd.api import Context
from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain
from ckb_zkp_amd.params import get_curve
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
c = get_curve("bn254")
ctx = Context(0)
inst = mimc_chain_instance(c, samples_for_domain(20))
params = groth16.generate_parameters(ctx, c, inst, alpha=11, beta=13, gamma=17, delta=19, tau=23)
pk = groth16.ProvingKey(ctx, params, inst)
z_dev = ctx.to_device(codec.fr_to_mont(inst.z, c).reshape(-1, 4))
rng = np.random.default_rng(7)
def rnd(k):
    return np.stack([codec.fr_to_mont([int.from_bytes(rng.bytes(32), "little") % c.r], c)[0] for _ in range(k)])
rw, sw = rnd(warm), rnd(warm)
rs = [(rnd(n), rnd(n)) for _ in range(reps)]
pk.prove_batch_raw([z_dev] * warm, rw, sw)
ctx.sync()
out = []
for r, s in rs:
    t0 = time.perf_counter()
    pk.prove_batch_raw([z_dev] * n, r, s)
    ctx.sync()
    out.append(round((time.perf_counter() - t0) * 1e3, 2))
print("warmup", warm, "n", n, "ms per batch:", out, "proofs/s:", [round(n * 1e3 / t, 1) for t in out])
