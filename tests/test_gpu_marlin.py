"""GPU: Marlin prover (BASELINE.json configs[3]) with fixed challenges — device-backed AHP rounds + KZG10 commitments,
evaluations and batch openings equal the big-int oracle's, and the resulting proof passes the reference's own
verifier (verifier_equality_check + PC::batch_check with a real pairing, marlin/src/lib.rs:184-250)."""
import random

import numpy as np
import pytest

from ckb_zkp_amd import codec, kzg10
from tests import marlin_hostlist as marlin
from ckb_zkp_amd.circuits import Mini, MimcChain
from ckb_zkp_amd.params import get_curve
from oracle.pyref import groth16 as og
from oracle.pyref import kzg10 as okzg
from oracle.pyref import marlin as om
from tests.util import OC

pytestmark = pytest.mark.gpu


def _rand_inputs(c, hs, seed):
    rnd = random.Random(seed)
    R = dict(w=[rnd.randrange(c.r)], z_a=[rnd.randrange(c.r)], z_b=[rnd.randrange(c.r)],
             mask=[rnd.randrange(c.r) for _ in range(3 * hs)],
             blind={l: [rnd.randrange(c.r), rnd.randrange(c.r)] for l in ("w", "z_a", "z_b", "g_1")},
             blind_shifted={"g_1": [rnd.randrange(c.r), rnd.randrange(c.r)]})
    ch = dict(alpha=rnd.randrange(c.r), eta_a=rnd.randrange(c.r), eta_b=rnd.randrange(c.r), eta_c=rnd.randrange(c.r),
              beta=rnd.randrange(c.r), gamma=rnd.randrange(c.r), xi=rnd.randrange(1 << 128))
    return R, ch


@pytest.mark.parametrize("curve,kind", [("bn254", "mimc"), ("bls12_381", "mimc"), ("bn254", "mini")])
def test_marlin_prover_matches_oracle_and_verifies(ctx, curve, kind):
    c = get_curve(curve)
    rnd = random.Random(5)
    if kind == "mimc":
        consts = [rnd.randrange(c.r) for _ in range(5)]
        pre = [(rnd.randrange(c.r), rnd.randrange(c.r)) for _ in range(2)]
        pcirc, pcirc_w = MimcChain(curve, consts, [(None, None)] * 2), MimcChain(curve, consts, pre)
        ocirc, public = og.MimcChain(OC[curve], consts, pre), []
    else:                                                   # marlin/tests/mini.rs: x*(y+2) = z, z public
        pcirc, pcirc_w = Mini(num=10), Mini(2, 3, 10, 10)
        ocirc, public = og.MiniCircuit(2, 3, 10, 10), [10]
    oidx = om.index(OC[curve], ocirc)
    idx = marlin.index(ctx, curve, pcirc)
    assert (idx["hs"], idx["ks"], idx["bs"], idx["xs"], idx["max_degree"]) == \
        (oidx["dh"].size, oidx["dk"].size, oidx["db"].size, oidx["dx"].size, oidx["max_degree"])
    for m in "abc":
        for k in ("row", "col", "val", "row_col"):
            assert idx["star"][m]["polys"][k] == oidx["star"][m]["polys"][k], (m, k)
            assert idx["star"][m]["on_b"][k] == oidx["star"][m]["on_b"][k], (m, k)
    beta_srs = 0x123456789ABCDEF
    pp = okzg.setup(OC[curve], idx["max_degree"], beta_srs)
    ck = kzg10.setup(ctx, curve, idx["max_degree"], beta_srs)
    try:
        R, ch = _rand_inputs(c, idx["hs"], seed=9)
        proof = marlin.create_proof(ctx, idx, ck, pcirc_w, R, ch)
        oproof = om.create_proof(oidx, pp, ocirc, R, ch)
        for l in marlin.LABELS_1 + marlin.LABELS_2 + marlin.LABELS_3:
            assert proof["polys"][l] == oproof["polys"][l], l
            assert proof["commitments"][l] == oproof["commitments"][l], l
        assert proof["query"] == oproof["query"] and proof["evaluations"] == oproof["evaluations"]
        assert proof["opening_proofs"] == oproof["opening_proofs"]
        # the reference's acceptance test on the DEVICE proof
        ic = om.index_commitments(oidx, pp)
        assert om.verify_proof(oidx, pp, ic, proof, public, ch)
        bad = dict(proof, evaluations=[(proof["evaluations"][0] + 1) % c.r] + proof["evaluations"][1:])
        assert not om.verify_proof(oidx, pp, ic, bad, public, ch)
    finally:
        ck.powers_of_g.free()
        ck.powers_of_gamma_g.free()


def test_device_vector_primitives(ctx):
    """spmv / gather / divide_by_vanishing / add-constant against plain big-int arithmetic."""
    from ckb_zkp_amd import marlin as marlin_native
    from tests import marlin_pyorch as marlin_dev
    c = get_curve("bn254")
    rnd = random.Random(3)
    be = marlin_dev.DeviceBackend(ctx, c)
    try:
        n, ncols = 300, 257
        x = [rnd.randrange(c.r) for _ in range(ncols)]
        rows = [[(rnd.choice([1, c.r - 1, rnd.randrange(c.r)]), rnd.randrange(ncols)) for _ in range(rnd.choice([0, 1, 3, 40]))]
                for _ in range(n)]
        # long rows: block-per-row partial sums (several 8192-term chunks) + per-row reduction
        rows[5] = [(rnd.choice([1, rnd.randrange(c.r)]), rnd.randrange(ncols)) for _ in range(20001)]
        rows[17] = [(rnd.randrange(c.r), rnd.randrange(ncols)) for _ in range(300)]
        rows[n - 1] = [(1, rnd.randrange(ncols)) for _ in range(8193)]
        got = be.download(be.spmv(marlin_dev._csr_dev(be, rows), be.upload(x), n))
        assert got == [sum(cf * x[j] for cf, j in row) % c.r for row in rows]
        idx = [rnd.choice([-1, rnd.randrange(ncols)]) for _ in range(500)]
        import numpy as np
        got = be.download(be.gather(be.upload(x), be.upload_raw(np.asarray(idx, dtype=np.int32)), len(idx)))
        assert got == [0 if j < 0 else x[j] for j in idx]
        for plen, dn in ((1000, 64), (64, 64), (50, 64), (129, 64), (777, 1), (5000, 1), (4097, 3), (100000, 2), (1, 1)):
            p = [rnd.randrange(c.r) for _ in range(plen)]
            q, rem = be.fold(be.upload(p), dn)
            eq, er = marlin.divide_by_vanishing(p, dn, c.r)
            assert marlin._trim(be.download(q)) == eq and marlin._trim(be.download(rem)) == er, (plen, dn)
        v = be.upload(x)
        assert be.download(be.addc(v, 12345)) == [(a + 12345) % c.r for a in x]
        be.add_at(v, 7, c.r - 5)
        assert be.element(v, 7) == (x[7] - 5) % c.r
        assert be.download(be.shift(v.view(0, 5), 3)) == [0, 0, 0] + x[:5]
    finally:
        be.release_all()


@pytest.mark.parametrize("curve,kind", [("bn254", "mimc"), ("bls12_381", "mimc"), ("bn254", "mini")])
def test_device_resident_marlin_matches_oracle_and_verifies(ctx, curve, kind):
    """marlin_dev.create_proof (all vectors resident in HBM) == the oracle prover, and the proof verifies."""
    from ckb_zkp_amd import marlin as marlin_native
    from tests import marlin_pyorch as marlin_dev
    c = get_curve(curve)
    rnd = random.Random(15)
    if kind == "mimc":
        consts = [rnd.randrange(c.r) for _ in range(5)]
        pre = [(rnd.randrange(c.r), rnd.randrange(c.r)) for _ in range(3)]
        pcirc, pcirc_w = MimcChain(curve, consts, [(None, None)] * 3), MimcChain(curve, consts, pre)
        ocirc, public = og.MimcChain(OC[curve], consts, pre), []
    else:
        pcirc, pcirc_w = Mini(num=10), Mini(2, 3, 10, 10)
        ocirc, public = og.MiniCircuit(2, 3, 10, 10), [10]
    oidx = om.index(OC[curve], ocirc)
    idx = marlin.index(ctx, curve, pcirc)
    didx = marlin_dev.DeviceIndex.from_host_index(ctx, idx)
    beta_srs = 0xFEDCBA987654321
    pp = okzg.setup(OC[curve], idx["max_degree"], beta_srs)
    ck = kzg10.setup(ctx, curve, idx["max_degree"], beta_srs)
    try:
        R, ch = _rand_inputs(c, idx["hs"], seed=21)
        timing = {}
        proof = marlin_dev.create_proof(ctx, didx, ck, pcirc_w, R, ch, timing)
        oproof = om.create_proof(oidx, pp, ocirc, R, ch)
        for l in marlin.LABELS_1 + marlin.LABELS_2 + marlin.LABELS_3:
            assert proof["commitments"][l] == oproof["commitments"][l], l
        assert proof["query"] == oproof["query"] and proof["evaluations"] == oproof["evaluations"]
        assert proof["opening_proofs"] == oproof["opening_proofs"]
        ic = om.index_commitments(oidx, pp)
        assert om.verify_proof(oidx, pp, ic, proof, public, ch)
        assert timing["total_s"] > 0
        # ---- create_random_proof: verifier messages DERIVED from the Fiat–Shamir transcript (lib.rs:105-158), by the
        # library's FiatShamirRng on the device side and by the oracle's independent implementation on the other
        assert didx.commit_index(ctx, ck) == ic
        ivk = marlin_native.index_verifier_key(didx, ck, ic, ck.vk_g2)
        assert ivk == om.index_verifier_key(oidx, pp, ic)
        rproof = marlin_dev.create_random_proof(ctx, didx, ck, ivk, pcirc_w, R)
        oproof = om.create_random_proof(oidx, pp, ic, ocirc, R)
        assert rproof["challenges"] == oproof["challenges"]
        assert rproof["commitments"] == oproof["commitments"] and rproof["evaluations"] == oproof["evaluations"]
        assert rproof["opening_proofs"] == oproof["opening_proofs"]
        wire = dict(commitments=rproof["commitments"], evaluations=rproof["evaluations"], opening_proofs=rproof["opening_proofs"])
        assert om.verify_random_proof(oidx, pp, ic, wire, public)          # marlin::verify_proof, challenges re-derived
        assert not om.verify_random_proof(oidx, pp, ic, wire, [(p + 1) % c.r for p in public] or [5])
        assert rproof["challenges"]["alpha"] != ch["alpha"]
    finally:
        ck.powers_of_g.free()
        ck.powers_of_gamma_g.free()


class _SwapAB:
    """the same circuit with the roles of A and B exchanged (A becomes the denser matrix: balance_matrices swaps rows)"""

    def __init__(self, inner):
        self.inner = inner

    def generate_constraints(self, cs):
        class Proxy:
            def __getattr__(self, name):
                return getattr(cs, name)

            def enforce(self, a, b, c):
                return cs.enforce(b, a, c)
        self.inner.generate_constraints(Proxy())


@pytest.mark.parametrize("curve,samples,swap", [("bn254", 3, False), ("bls12_381", 2, False), ("bn254", 40, False),
                                                ("bn254", 7, True)])
def test_device_index_from_arrays_matches_host_index(ctx, curve, samples, swap):
    """DeviceIndex.from_instance (arithmetization computed on the device from CSR arrays) == marlin.index on the same
    circuit synthesised constraint by constraint; the array-form prover input gives the same proof."""
    from ckb_zkp_amd import marlin as marlin_native
    from tests import marlin_pyorch as marlin_dev
    from ckb_zkp_amd.circuits import mimc_chain_instance
    c = get_curve(curve)
    inst = mimc_chain_instance(curve, samples, seed=77)
    circ = MimcChain(curve, inst.constants, inst.preimages)
    icirc = MimcChain(curve, inst.constants, [(None, None)] * samples)
    if swap:
        from ckb_zkp_amd.r1cs import R1csInstance
        inst = R1csInstance(curve, inst.num_inputs, inst.num_aux, inst.num_constraints(), inst.csr("b"), inst.csr("a"),
                            inst.csr("c"), inst.z)
        circ, icirc = _SwapAB(circ), _SwapAB(icirc)
    idx = marlin.index(ctx, curve, icirc)
    d_host = marlin_dev.DeviceIndex.from_host_index(ctx, idx)
    d_arr = marlin_dev.DeviceIndex.from_instance(ctx, inst)
    assert (d_arr.xs, d_arr.hs, d_arr.ks, d_arr.bs, d_arr.max_degree, d_arr.nrows) == \
        (idx["xs"], idx["hs"], idx["ks"], idx["bs"], idx["max_degree"], len(idx["a"]))
    be = d_arr.be
    for l in marlin.INDEX_LABELS:
        m, k = l.split("_", 1)
        assert be.download(d_arr.polys[l]) == idx["star"][m]["polys"][k], l
        assert be.download(d_arr.on_b[m][k]) == idx["star"][m]["on_b"][k], l
    ck = kzg10.setup(ctx, curve, idx["max_degree"], 0xABCDEF12345)
    try:
        R, ch = _rand_inputs(c, idx["hs"], seed=33)
        p_host = marlin_dev.create_proof(ctx, d_host, ck, circ, R, ch)
        p_arr = marlin_dev.create_proof(ctx, d_arr, ck, (inst.z[:1], inst.z[1:]), R, ch)
        assert p_host == p_arr
        p_ref = marlin.create_proof(ctx, idx, ck, circ, R, ch)
        assert p_ref["commitments"] == p_arr["commitments"] and p_ref["evaluations"] == p_arr["evaluations"]
        assert p_ref["opening_proofs"] == p_arr["opening_proofs"]
    finally:
        ck.powers_of_g.free()
        ck.powers_of_gamma_g.free()


def test_marlin_config4_full_size_verifies(ctx):
    """BASELINE.json configs[3] at full size (|H| = 2^20, |K| = 2^21, |B| = 2^23, SRS degree 6.29 M): the device-resident
    prover's proof passes the reference's verifier (AHP equality checks + KZG10 pairing checks, lib.rs:184-250) against
    index commitments computed on the device, and a tampered evaluation is rejected — the size-independent acceptance
    test the reference itself uses (marlin/tests/mini.rs:81-87)."""
    from ckb_zkp_amd import codec
    from ckb_zkp_amd import marlin as marlin_native
    from tests import marlin_pyorch as marlin_dev
    from ckb_zkp_amd.circuits import mimc_chain_instance
    from oracle.pyref.curves import Group
    from oracle.pyref.ntt import Domain
    curve = "bn254"
    c = get_curve(curve)
    from tests.util import TEST_FULL
    lh = 20 if TEST_FULL else 18                                  # ZKP_TEST_FULL=0: |H| = 2^18, same steps
    inst = mimc_chain_instance(curve, 87381 if TEST_FULL else 21845, seed=0x4D41)
    didx = marlin_dev.DeviceIndex.from_instance(ctx, inst)
    assert (didx.hs, didx.ks, didx.bs) == (1 << lh, 2 << lh, 8 << lh)
    beta_srs = 0x0F1E2D3C4B5A69788796A5B4C3D2E1F0
    ck = kzg10.setup(ctx, curve, didx.max_degree, beta_srs)
    try:
        rnd = random.Random(44)
        R, ch = _rand_inputs(c, 1, seed=45)
        mask_ints = [rnd.randrange(c.r) for _ in range(3 * didx.hs)]
        R["mask"] = codec.fr_to_mont(mask_ints, c).reshape(-1, 4)
        oc = OC[curve]
        G1, G2 = Group(oc, 1), Group(oc, 2)
        pp = dict(curve=oc, g=G1.gen, gamma_g=G1.mul(G1.gen, 7), h=G2.gen, beta_h=G2.mul(G2.gen, beta_srs))
        assert ck.vk_g2 == (pp["h"], pp["beta_h"])
        oidx = dict(curve=oc, dh=Domain(oc, didx.hs), dk=Domain(oc, didx.ks), max_degree=didx.max_degree,
                    num_variables=didx.nrows, num_constraints=didx.nrows, num_non_zeros=didx.num_non_zeros)
        ic = didx.commit_index(ctx, ck)
        # create_random_proof: every verifier message derived from the transcript; the oracle's verifier re-derives them
        ivk = marlin_native.index_verifier_key(didx, ck, ic, ck.vk_g2)
        proof = marlin_dev.create_random_proof(ctx, didx, ck, ivk, (inst.z[:1], inst.z[1:]), R)
        wire = dict(commitments=proof["commitments"], evaluations=proof["evaluations"], opening_proofs=proof["opening_proofs"])
        assert om.verify_random_proof(oidx, pp, ic, wire, [])
        bad = dict(wire, evaluations=wire["evaluations"][:3] + [(wire["evaluations"][3] + 1) % c.r] + wire["evaluations"][4:])
        assert not om.verify_random_proof(oidx, pp, ic, bad, [])
        # and the fixed-challenge test hook still yields an accepting proof for the supplied messages
        proof2 = marlin_dev.create_proof(ctx, didx, ck, (inst.z[:1], inst.z[1:]), R, ch)
        assert om.verify_proof(oidx, pp, ic, proof2, [], ch)
        # the C entry point (zkp_marlin_index_upload + zkp_marlin_prove) at full size: same index commitments, the same
        # proof as the Python-orchestrated device prover (same randomness, same transcript), accepted by the verifier
        nidx = marlin_native.NativeIndex(ctx, inst)
        try:
            assert nidx.commit_index(ck) == ic
            nproof = marlin_native.prove_native(ctx, nidx, ck, ivk, inst.z[:1], inst.z[1:], R)
            assert nproof["challenges"] == proof["challenges"]
            assert nproof["commitments"] == proof["commitments"] and nproof["evaluations"] == proof["evaluations"]
            assert nproof["opening_proofs"] == proof["opening_proofs"]
            # ORACLE at full size (VERDICT r4 task 1): marlin::create_random_proof restated in C++ (oracle/cpu/marlin_oracle.inc,
            # == oracle/pyref/marlin.py on the small circuits of tests/test_oracle_marlin_cpu.py), run WITHOUT any product code:
            # index from the instance as synthesised, transcript by oracle/pyref/fs_rng.py, the committer key's host arrays as the
            # only shared input (spot-checked below against beta^i * g with the oracle's own scalar multiplication).
            from oracle import cpu_oracle
            co = cpu_oracle.MarlinOracle(oc, inst, srs=(ck.host_g, ck.host_gamma_g))
            try:
                assert (co.xs, co.hs, co.ks, co.bs, co.max_degree, co.num_non_zeros) == \
                    (nidx.xs, nidx.hs, nidx.ks, nidx.bs, nidx.max_degree, nidx.num_non_zeros)
                for i in (0, 1, 2, 12345, didx.hs, didx.max_degree):
                    want_pt = G1.mul(G1.gen, pow(beta_srs, i, oc.r))
                    assert codec.g1_from_mont(ck.host_g[0][i:i + 1], ck.host_g[1][i:i + 1], c)[0] == want_pt, i
                assert codec.g1_from_mont(ck.host_gamma_g[0][1:2], ck.host_gamma_g[1][1:2], c)[0] == G1.mul(G1.gen, 7 * beta_srs % oc.r)
                assert co.index_commitments() == ic
                o = co.create_proof(inst.z[:1], codec.fr_to_mont(inst.z[1:], c).reshape(-1, 4), R,
                                    om.FiatShamirChallenger(oidx, ivk, []))
                assert o["challenges"] == nproof["challenges"]
                for l in marlin.LABELS_1 + marlin.LABELS_2 + marlin.LABELS_3:
                    assert nproof["commitments"][l] == o["commitments"][l], l          # 9 commitments + 2 shifted
                assert nproof["query"] == o["query"]
                assert nproof["evaluations"] == o["evaluations"]                      # all 21
                assert nproof["opening_proofs"] == o["opening_proofs"]                # both (w, rand_v)
            finally:
                co.free()
        finally:
            nidx.free()
    finally:
        ck.powers_of_g.free()
        ck.powers_of_gamma_g.free()
        marlin_dev.DeviceBackend.trim_pool(ctx)
        didx.free()


@pytest.mark.parametrize("curve,samples,swap", [("bn254", 3, False), ("bls12_381", 2, False), ("bn254", 7, True)])
def test_native_marlin_prover_matches_oracle(ctx, curve, samples, swap):
    """zkp_marlin_index_upload / zkp_marlin_index_commit / zkp_marlin_prove (csrc/marlin.hip: the whole create_random_proof
    behind the C ABI, no Python between the rounds) against the oracle: index commitments, the proof for supplied verifier
    messages (test hook) and the proof with messages DERIVED from the Fiat–Shamir transcript are bit-identical to the
    oracle's, and the oracle's verifier (which re-derives the messages) accepts; a wrong public input is rejected."""
    from ckb_zkp_amd import marlin as marlin_native
    from tests import marlin_pyorch as marlin_dev
    from ckb_zkp_amd.circuits import mimc_chain_instance
    c = get_curve(curve)
    inst = mimc_chain_instance(curve, samples, seed=91)
    ocirc = og.MimcChain(OC[curve], inst.constants, inst.preimages)
    if swap:
        from ckb_zkp_amd.r1cs import R1csInstance
        inst = R1csInstance(curve, inst.num_inputs, inst.num_aux, inst.num_constraints(), inst.csr("b"), inst.csr("a"),
                            inst.csr("c"), inst.z)
        ocirc = _SwapAB(ocirc)
    oidx = om.index(OC[curve], ocirc)
    nidx = marlin_native.NativeIndex(ctx, inst)
    assert (nidx.xs, nidx.hs, nidx.ks, nidx.bs, nidx.max_degree, nidx.num_non_zeros) == \
        (oidx["dx"].size, oidx["dh"].size, oidx["dk"].size, oidx["db"].size, oidx["max_degree"], oidx["num_non_zeros"])
    beta_srs = 0x13579BDF2468ACE
    pp = okzg.setup(OC[curve], nidx.max_degree, beta_srs)
    ck = kzg10.setup(ctx, curve, nidx.max_degree, beta_srs)
    try:
        ic = om.index_commitments(oidx, pp)
        assert nidx.commit_index(ck) == ic
        R, ch = _rand_inputs(c, nidx.hs, seed=57)
        x, w = inst.z[:inst.num_inputs], inst.z[inst.num_inputs:]
        # supplied verifier messages (test hook)
        p = marlin_native.prove_native(ctx, nidx, ck, None, x, w, R, ch)
        o = om.create_proof(oidx, pp, ocirc, R, ch)
        assert p["commitments"] == o["commitments"]
        assert p["query"] == o["query"] and p["evaluations"] == o["evaluations"]
        assert p["opening_proofs"] == o["opening_proofs"]
        # create_random_proof: messages derived inside the library
        ivk = om.index_verifier_key(oidx, pp, ic)
        assert ck.vk_g2 == (pp["h"], pp["beta_h"])
        p = marlin_native.prove_native(ctx, nidx, ck, ivk, x, w, R)
        o = om.create_random_proof(oidx, pp, ic, ocirc, R)
        assert p["challenges"] == o["challenges"]
        assert p["commitments"] == o["commitments"] and p["evaluations"] == o["evaluations"]
        assert p["opening_proofs"] == o["opening_proofs"]
        wire = dict(commitments=p["commitments"], evaluations=p["evaluations"], opening_proofs=p["opening_proofs"])
        assert om.verify_random_proof(oidx, pp, ic, wire, x[1:])
        assert not om.verify_random_proof(oidx, pp, ic, wire, [(v + 1) % c.r for v in x[1:]] or [3])
        # a second proof on the same index (pooled scratch reused) with other randomness differs and verifies
        R2, _ = _rand_inputs(c, nidx.hs, seed=58)
        p2 = marlin_native.prove_native(ctx, nidx, ck, ivk, x, w, R2)
        assert p2["commitments"] != p["commitments"]
        wire2 = dict(commitments=p2["commitments"], evaluations=p2["evaluations"], opening_proofs=p2["opening_proofs"])
        assert om.verify_random_proof(oidx, pp, ic, wire2, x[1:])
    finally:
        nidx.free()
        ck.powers_of_g.free()
        ck.powers_of_gamma_g.free()
