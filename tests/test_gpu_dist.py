"""GPU: the multi-GPU data path (BASELINE.json configs[4]) on one device — both ranks of a world_size-2 job are
simulated in-process (no collective), so what is checked is the device side: sliced uploads, partial MSMs on
Montgomery scalars (G1 and G2), zkp_g*_fold, zkp_groth16_assemble.  The collective itself is covered on CPU by
tests/test_dist_gloo.py."""
import os

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


def _device_sharded_proof(ctx, params, inst, z_dev, world, r_, s_):
    """all ranks of the device-resident sharded step simulated in one process: rank k's partial sums land in slot k of
    the buffer the all-gather would fill; fold + assembly read it in place (no host copy of any point)."""
    pb = groth16.partials_bytes(ctx, params.curve)
    gathered = ctx.dev_alloc(world * pb)
    try:
        for rank in range(world):
            pk_s = groth16.ProvingKey(ctx, params, inst, shard=(rank, world))
            try:
                pk_s.partials_dev(z_dev, r_, s_, gathered + rank * pb)
                from ckb_zkp_amd._lib import ZkpError
                if rank == 0:
                    c = params.curve
                    with pytest.raises(ZkpError):        # a sharded key yields partial sums only
                        pk_s.prove_raw(z_dev, codec.fr_to_mont([r_], c)[0], codec.fr_to_mont([s_], c)[0], z_on_device=True)
            finally:
                pk_s.free()
        return groth16.fold_assemble_dev(ctx, params.curve, gathered, world, r_, s_)
    finally:
        ctx.dev_free(gathered)


@pytest.mark.parametrize("curve,k,world", [("bn254", 12, 2), ("bn254", 11, 8), ("bn254", 4, 8), ("bls12_381", 9, 3)])
def test_device_sharded_step_equals_single_gpu_proof(ctx, curve, k, world):
    """zkp_groth16_pk_upload_shard / zkp_groth16_prove_partials_dev / zkp_groth16_fold_assemble_dev: the proof of the
    base-sharded step (world ranks, slices that do not divide evenly, k = 4: ranks with EMPTY slices) is bit-identical
    to the single-GPU proof, for (r, s) random and (0, 0)."""
    c = get_curve(curve)
    inst = mimc_chain_instance(curve, samples_for_domain(k))
    params = groth16.generate_parameters(ctx, curve, inst, **TOXIC)
    pk = groth16.ProvingKey(ctx, params, inst)
    z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
    zd = ctx.to_device(z)
    try:
        for r_, s_ in ((0xABCDEF0123456789ABCDEF, 0x13579BDF02468ACE), (0, 0)):
            out1, inf1 = pk.prove_raw(zd, codec.fr_to_mont([r_], c)[0], codec.fr_to_mont([s_], c)[0], z_on_device=True)
            out2, inf2 = _device_sharded_proof(ctx, params, inst, zd, world, r_, s_)
            assert np.array_equal(out1, out2) and np.array_equal(inf1, inf2), (r_, s_)
    finally:
        ctx.dev_free(zd)
        pk.free()


def test_config5_2p24_single_gpu_and_8way_sharded(ctx):
    """BASELINE.json configs[4] at FULL size: 16 777 210-constraint MiMC chain, domain 2^24, BN254.
    (1) the single-GPU proof equals the proof computed in the exponent from the toxic waste (size-independent check,
    SURVEY §8(c).3) and h has degree <= N-2; (2) the 8-way base-sharded device-resident step (every rank's slice of the
    20 132 656-point queries, partial sums all-gathered in HBM, folded and assembled on the device) returns the same
    proof bit for bit (through the in-library multi-GPU path by default, ZKP_TEST_2P24_SHARD=step for the per-process step:
    tests at <= 2^12 above cover that one on every run).  ~100 GB of HBM, a few minutes."""
    from oracle.pyref.curves import Group
    from tests.util import OC
    from tests.util import TEST_FULL
    curve, k, world = "bn254", 24 if TEST_FULL else 22, 8       # ZKP_TEST_FULL=0: the same steps at 2^22
    import time
    tt = [time.time()]

    def lap(what):                                              # (pytest -s: where the minutes of this test go)
        tt.append(time.time())
        print(f"[2^{k}] {what}: {tt[-1] - tt[-2]:.1f} s")
    c = get_curve(curve)
    inst = mimc_chain_instance(curve, samples_for_domain(k))
    lap("instance")
    params = groth16.generate_parameters(ctx, curve, inst, **TOXIC)
    lap("synthetic key")
    z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
    zd = ctx.to_device(z)
    r_, s_ = 0x1F2E3D4C5B6A7988, 0x8899AABBCCDDEEFF
    try:
        pk = groth16.ProvingKey(ctx, params, inst)
        lap("key upload")
        try:
            assert pk.domain_size == 1 << k
            from oracle import cpu_oracle
            thr = cpu_oracle.hardware_threads()
            h_dev = pk.witness_map(z)
            rm, sm = codec.fr_to_mont([r_], c)[0], codec.fr_to_mont([s_], c)[0]
            out1, inf1 = pk.prove_raw(zd, rm, sm, z_on_device=True)
            # full-size pin at 2^24: h and the whole proof equal the C++ restatement of the reference (all host threads; ONE oracle
            # pass returns the proof and the quotient it was made from)
            lap("device witness map + proof")
            o_out, o_inf, _, o_h = cpu_oracle.groth16_prove(params, inst, z, rm, sm, threads=thr, want_h=True)
            lap("oracle/cpu proof")
            assert np.array_equal(h_dev, o_h)
            del o_h
            assert not h_dev[-1].any()                          # deg h <= N - 2
            assert np.array_equal(out1, o_out) and np.array_equal(inf1, o_inf)
            proof = pk.decode_proof(out1, inf1)
        finally:
            pk.free()
        # the key's exponents (computed with the library's Fr kernels at this size) re-derived for a sample with Python integers,
        # then the proof in the exponent (inner products over the full assignment by oracle/cpu)
        from tests.util import spot_check_qap_exponents, trapdoor_proof_exponents
        spot_check_qap_exponents(params, inst)
        lap("exponent spot check")
        A, B, Cc = trapdoor_proof_exponents(params, inst, z, h_dev, r_, s_)
        del h_dev
        lap("trapdoor exponents")
        G1, G2 = Group(OC[curve], 1), Group(OC[curve], 2)
        assert proof.a == G1.mul(G1.gen, A)
        assert proof.b == G2.mul(G2.gen, B)
        assert proof.c == G1.mul(G1.gen, Cc)
        import time
        t0 = time.time()
        if os.environ.get("ZKP_TEST_2P24_SHARD", "multi") == "step":
            # one process per GPU style: zkp_groth16_pk_upload_shard x 8 (every rank transforms the H query itself: ~2 min here)
            out2, inf2 = _device_sharded_proof(ctx, params, inst, zd, world, r_, s_)
        else:
            # ONE process, eight ranks: zkp_ctx_create_multi + zkp_groth16_pk_upload_multi(SHARD) + zkp_groth16_prove_multi (the path of
            # `bench.py --gpus 8 --mode shard`; the H query is transformed once and sliced) — witness on the host and per-rank on the device
            from ckb_zkp_amd.api import MultiContext
            m = MultiContext([ctx.device] * world)
            try:
                mpk = groth16.MultiProvingKey(m, params, inst, groth16.MULTI_SHARD)
                try:
                    out2, inf2 = mpk.prove_raw(z, rm, sm)
                    zds = [m.member(k_).to_device(z) for k_ in range(world)]
                    out3, inf3 = mpk.prove_raw(zds, rm, sm, z_on_device=True)
                    for k_, d_ in enumerate(zds):
                        m.member(k_).dev_free(d_)
                    assert np.array_equal(out1, out3) and np.array_equal(inf1, inf3)
                    assert mpk.info()["devices"] == world
                finally:
                    mpk.free()
            finally:
                m.close()
        print(f"8-way sharded leg: {time.time() - t0:.1f} s")
        assert np.array_equal(out1, out2) and np.array_equal(inf1, inf2)
    finally:
        ctx.dev_free(zd)
