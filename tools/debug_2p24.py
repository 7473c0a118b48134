#!/usr/bin/env python3
"""debug helper: one 2^k BN254 proof through zkp_groth16_prove_dev with serialised kernels"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ckb_zkp_amd import codec, groth16
from ckb_zkp_amd.api import Context
from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain
k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
TOXIC = dict(alpha=0x1234567890ABCDEF1, beta=0xFEDCBA09876543211, gamma=0x1111111111111111111, delta=0x2222222222222222223, tau=0x3333333333333333335)
ctx = Context(0)
inst = mimc_chain_instance("bn254", samples_for_domain(k))
print("instance", flush=True)
params = groth16.generate_parameters(ctx, "bn254", inst, **TOXIC)
print("params", flush=True)
pk = groth16.ProvingKey(ctx, params, inst)
print("pk", flush=True)
c = params.curve
z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
zd = ctx.to_device(z)
prof = len(sys.argv) > 2
if prof:
    ctx.set_profiling(True)
for i in range(2):
    t = time.time()
    out, inf = pk.prove_raw(zd, codec.fr_to_mont([5], c)[0], codec.fr_to_mont([7], c)[0], z_on_device=True)
    print("proof", i, time.time() - t, flush=True)
    if prof:
        print(pk.last_timing(), flush=True)
