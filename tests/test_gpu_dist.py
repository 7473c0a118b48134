"""GPU: the multi-GPU data path (BASELINE.json configs[4]) on one device — both ranks of a world_size-2 job are
simulated in-process (no collective), so what is checked is the device side: sliced uploads, partial MSMs on
Montgomery scalars (G1 and G2), zkp_g*_fold, zkp_groth16_assemble.  The collective itself is covered on CPU by
tests/test_dist_gloo.py."""
import numpy as np
import pytest

from ckb_zkp_amd import codec, groth16
from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain
from ckb_zkp_amd.distributed import GpuEngine, ShardedBases, ShardedGroth16Prover
from ckb_zkp_amd.params import get_curve

pytestmark = pytest.mark.gpu
TOXIC = dict(alpha=0x1234567890ABCDEF1, beta=0xFEDCBA09876543211, gamma=0x1111111111111111111,
             delta=0x2222222222222222223, tau=0x3333333333333333335)


@pytest.mark.parametrize("curve,k,world", [("bn254", 12, 2), ("bn254", 10, 3), ("bls12_381", 9, 2)])
def test_sharded_prover_equals_single_gpu_prover(ctx, curve, k, world):
    c = get_curve(curve)
    inst = mimc_chain_instance(curve, samples_for_domain(k))
    params = groth16.generate_parameters(ctx, curve, inst, **TOXIC)
    pk = groth16.ProvingKey(ctx, params, inst)
    pk_m = groth16.ProvingKey(ctx, params, inst, matrices_only=True)
    try:
        z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
        r_, s_ = 0xABCDEF0123456789ABCDEF, 0x13579BDF02468ACE
        out1, inf1 = pk.prove_raw(z, codec.fr_to_mont([r_], c)[0], codec.fr_to_mont([s_], c)[0])
        eng = GpuEngine(ctx)
        provers = [ShardedGroth16Prover(eng, params, inst, rank, world, witness_mapper=pk_m.witness_map)
                   for rank in range(world)]
        h = pk_m.witness_map(z)
        parts = np.stack([p.partial_sums(z, h, r_, s_) for p in provers])       # what the all-gather delivers
        sums = provers[0].fold_sums(parts)
        out2, inf2 = groth16.assemble(ctx, c, sums, r_, s_)
        assert np.array_equal(out1, out2) and np.array_equal(inf1, inf2)
        # device-resident step (world 1: the whole path on one rank, no host scalars): same proof
        solo = ShardedGroth16Prover(eng, params, inst, 0, 1, witness_mapper=pk_m.witness_map)
        zd = ctx.to_device(z)
        out3, inf3 = groth16.assemble(ctx, c, solo.prove_sums_dev(ctx, pk_m, zd, r_, s_), r_, s_)
        assert np.array_equal(out1, out3) and np.array_equal(inf1, inf3)
        # ... and the partial sums of the sharded ranks computed from device-resident scalars
        S = ctx.to_device(np.concatenate([z, codec.fr_to_mont([1, r_, s_, (-(r_ * s_)) % c.r], c).reshape(4, 4)]))
        nz = z.shape[0]
        for pr, want in zip(provers, parts):
            got = pr.A.partial_dev(S, nz + 4)
            wa = want[:got.shape[0]]
            assert ctx.into_affine(c, 1, got)[0].tolist() == ctx.into_affine(c, 1, wa)[0].tolist()
            f = c.fq_limbs
            got2 = pr.B2.partial_dev(S, nz + 4)
            assert ctx.into_affine(c, 2, got2)[0].tolist() == ctx.into_affine(c, 2, want[6 * f:12 * f])[0].tolist()
        ctx.dev_free(zd)
        ctx.dev_free(S)
        # matrices-only key refuses to prove
        from ckb_zkp_amd._lib import ZkpError
        with pytest.raises(ZkpError):
            pk_m.prove_raw(z, codec.fr_to_mont([r_], c)[0], codec.fr_to_mont([s_], c)[0])
    finally:
        pk.free()
        pk_m.free()
