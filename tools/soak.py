#!/usr/bin/env python3
"""Soak of the pipelined prover: many batches of varying length, two witnesses, random (r, s); every proof of every batch must equal the
blocking single-proof call with the same inputs (lanes, streams, bucket chaining, host-side into_affine under load).
    python tools/soak.py [curve=bn254] [log_n=16] [proofs=3000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ckb_zkp_amd import codec, groth16
from ckb_zkp_amd.api import Context
from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain
from ckb_zkp_amd.params import get_curve
c = get_curve(sys.argv[1] if len(sys.argv) > 1 else "bn254")
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
total = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
ctx = Context(0)
S = samples_for_domain(k)
insts = [mimc_chain_instance(c, S, seed=s) for s in (1, 2)]
params = groth16.generate_parameters(ctx, c, insts[0], alpha=11, beta=12, gamma=13, delta=14, tau=987654321)
pk = groth16.ProvingKey(ctx, params, insts[0])
zs = [codec.fr_to_mont(i.z, c).reshape(-1, 4) for i in insts]
zd = [ctx.to_device(z) for z in zs]
rng = np.random.default_rng(99)
def fr(n):
    return codec.fr_to_mont([int.from_bytes(rng.bytes(40), "little") % c.r for _ in range(n)], c).reshape(-1, 4)
# reference answers from blocking calls for a pool of (witness, r, s)
pool = 64
R, Sx = fr(pool), fr(pool)
W = rng.integers(0, 2, pool)
ref = [pk.prove_raw(zd[W[i]], R[i], Sx[i], z_on_device=True) for i in range(pool)]
done, bad, t0 = 0, 0, time.time()
while done < total:
    n = int(rng.choice([1, 2, 3, 7, 8, 9, 16, 33, 64, 100]))
    idx = rng.integers(0, pool, n)
    outs, infs = pk.prove_batch_raw([zd[W[i]] for i in idx], R[idx], Sx[idx])
    for j, i in enumerate(idx):
        if not (np.array_equal(outs[j], ref[i][0]) and np.array_equal(infs[j], ref[i][1])):
            bad += 1
    done += n
print(f"soak {c.name} 2^{k}: {done} pipelined proofs in {time.time() - t0:.1f} s, mismatches vs the blocking call: {bad}")
sys.exit(1 if bad else 0)
