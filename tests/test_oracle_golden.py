"""CPU: both oracles (pyref big-int, C++ restatement) against the committed golden vectors, and the oracle's
Groth16 proofs against the REFERENCE'S OWN acceptance test (pairing verification, groth16/src/verifier.rs:18-44,
as exercised by groth16/tests/mini.rs:89,96)."""
import numpy as np
import pytest

from ckb_zkp_amd import codec
from ckb_zkp_amd.params import get_curve
from ckb_zkp_amd.r1cs import ConstraintSystem, R1csInstance
from oracle import cpu_oracle
from oracle.pyref import fields as ofields
from oracle.pyref import groth16 as og
from oracle.pyref.curves import Group, ark_window_bits
from oracle.pyref.ntt import Domain
from oracle.pyref.pairing import verify_proof
from tests.golden_util import GOLDEN, I, TOXIC, abi_params_from_oracle, golden_circuits, unpt
from tests.util import OC, jac_limbs_to_affine_oracle, to_abi_points

CURVES = ["bn254", "bls12_381"]
OPS = ["fft", "ifft", "coset_fft", "coset_ifft"]


def test_constants_rederived():
    ofields.self_check()
    assert [ark_window_bits(n) for n in (31, 32, 1 << 20, (1 << 20) - 1, 1258284, 1 << 22, 1 << 24)] == \
        [3, 5, 15, 15, 16, 17, 18]                      # BASELINE.md §2 / SURVEY §2.2


def test_public_known_answers_pin_the_curve_arithmetic():
    """Values published OUTSIDE this repository and the reference (the reference's tests hold no vectors for this path): the
    alt_bn128 / BN254 moduli, generators and 2*G1 of EIP-196 / EIP-197, the BLS12-381 moduli and G1 generator of the IETF
    pairing-friendly-curves draft.  They pin the oracle's field and group arithmetic; everything else is checked against it."""
    from oracle.pyref.curves import Group
    bn, bls = OC["bn254"], OC["bls12_381"]
    assert bn.q == 21888242871839275222246405745257275088696311157297823662689037894645226208583
    assert bn.r == 21888242871839275222246405745257275088548364400416034343698204186575808495617
    assert tuple(bn.g1_gen) == (1, 2)
    g1 = Group(bn, 1)
    two_g = g1.add(g1.gen, g1.gen)                                   # EIP-196 test vector (ecAdd of the generator with itself)
    assert two_g == (0x030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3,
                     0x15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4)
    assert g1.mul(g1.gen, 2) == two_g and g1.mul(g1.gen, bn.r) is None and g1.mul(g1.gen, bn.r + 2) == two_g
    assert (tuple(bn.g2_gen[0]), tuple(bn.g2_gen[1])) == (                                               # EIP-197
        (10857046999023057135944570762232829481370756359578518086990519993285655852781,
         11559732032986387107991004021392285783925812861821192530917403151452391805634),
        (8495653923123431417604973247489272438418190587263600148770280649306958101930,
         4082367875863433681332203403145435568316851327593401208105741076214120093531))
    g2 = Group(bn, 2)
    assert g2.on_curve(g2.gen) and g2.mul(g2.gen, bn.r) is None
    assert bls.q == 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    assert bls.r == 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    assert tuple(bls.g1_gen) == (
        0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
        0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)
    for grp in (Group(bls, 1), Group(bls, 2)):
        assert grp.on_curve(grp.gen) and grp.mul(grp.gen, bls.r) is None


@pytest.mark.parametrize("curve", CURVES)
def test_ntt_golden(curve):
    c = get_curve(curve)
    for e in GOLDEN["curves"][curve]["ntt"]:
        x = [I(v) for v in e["input"]]
        d = Domain(OC[curve], e["n"])
        assert d.fft(x) == d.dft_naive(x) and d.ifft(x) == d.dft_naive(x, inverse=True)   # O(n^2) definition
        for op, name in enumerate(OPS):
            exp = [I(v) for v in e[name]]
            assert getattr(d, name)(x) == exp
            for th in (1, 4):
                got = codec.fr_from_mont(cpu_oracle.ntt(c.cid, codec.fr_to_mont(x, c), op, threads=th), c)
                assert got == exp, (curve, name, th)


@pytest.mark.parametrize("curve", CURVES)
def test_msm_golden(curve):
    c = get_curve(curve)
    for e in GOLDEN["curves"][curve]["msm"]:
        g = e["group"]
        G = Group(OC[curve], g)
        pts = [unpt(p, g) for p in e["bases"]]
        ks = [I(k) for k in e["scalars"]]
        exp = unpt(e["result"], g)
        assert all(G.on_curve(p) for p in pts)
        assert G.msm_naive(pts, ks) == exp and G.msm_pippenger(pts, ks) == exp
        xy, inf = to_abi_points(curve, g, pts)
        for th in (1, 3):
            out = cpu_oracle.msm(c.cid, g, xy, inf, codec.fr_canonical(ks, c), threads=th)
            assert jac_limbs_to_affine_oracle(curve, g, out) == exp
        # ark min(len) truncation
        out = cpu_oracle.msm(c.cid, g, xy, inf, codec.fr_canonical(ks[:5], c))
        assert jac_limbs_to_affine_oracle(curve, g, out) == G.msm_naive(pts[:5], ks[:5])


@pytest.mark.parametrize("curve", CURVES)
def test_groth16_golden_and_reference_verifier(curve):
    """Oracle prover reproduces the golden proofs; the proofs pass `verify_proof`; a tampered proof and a wrong
    public input fail; the C++ restatement of create_proof gives the same three points and the same h."""
    c = get_curve(curve)
    for e in GOLDEN["curves"][curve]["groth16"]:
        ocirc, ocirc_setup, pcirc, pcirc_setup = golden_circuits(curve, e)
        opk = og.generate_parameters(OC[curve], ocirc_setup, **TOXIC, g1_k=e["g1_k"], g2_k=e["g2_k"])
        r_, s_ = I(e["r"]), I(e["s"])
        proof, inter = og.create_proof(opk, ocirc, r_, s_, msm="pippenger")
        exp = (unpt(e["a"], 1), unpt(e["b"], 2), unpt(e["c"], 1))
        assert (proof.a, proof.b, proof.c) == exp
        assert inter["h"] == [I(v) for v in e["h"]]
        pub = [10] if e["circuit"] == "mini" else []
        assert verify_proof(OC[curve], opk, proof, pub)
        if e["circuit"] == "mini":
            assert not verify_proof(OC[curve], opk, proof, [11])
            G1 = Group(OC[curve], 1)
            assert not verify_proof(OC[curve], opk, og.Proof(proof.a, proof.b, G1.add(proof.c, G1.gen)), pub)
        # C++ restatement through the same descriptor struct the product ABI uses
        cs = ConstraintSystem(curve, True)
        pcirc.generate_constraints(cs)
        inst = R1csInstance.from_cs(cs)
        params = abi_params_from_oracle(curve, opk, inst.num_inputs, inst.num_aux, inst.num_constraints())
        z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
        assert codec.fr_from_mont(cpu_oracle.witness_map(params, inst, z, threads=2), c) == inter["h"]
        out, inf, _ = cpu_oracle.groth16_prove(params, inst, z, codec.fr_to_mont([r_], c)[0],
                                               codec.fr_to_mont([s_], c)[0], threads=3)
        f = c.fq_limbs
        got = (codec.g1_from_mont(out[:2 * f], [inf[0]], c)[0], codec.g2_from_mont(out[2 * f:6 * f], [inf[1]], c)[0],
               codec.g1_from_mont(out[6 * f:], [inf[2]], c)[0])
        assert got == exp


def test_cpp_oracle_midsize_vs_python_pippenger():
    """2^10-point MSM + 2^12 NTT: C++ restatement == Python restatement (different code, same algorithm)."""
    import random
    c = get_curve("bn254")
    G = Group(OC["bn254"], 1)
    rnd = random.Random(5)
    n = 1 << 10
    ds = [rnd.randrange(1, c.r) for _ in range(64)]
    base = [G.mul(G.gen, d) for d in ds]
    pts = [base[i % 64] for i in range(n)]
    ks = [rnd.randrange(c.r) for _ in range(n)]
    xy, inf = to_abi_points("bn254", 1, pts)
    out = cpu_oracle.msm(0, 1, xy, inf, codec.fr_canonical(ks, c), threads=4)
    e = sum(ds[i % 64] * k for i, k in enumerate(ks)) % c.r
    assert jac_limbs_to_affine_oracle("bn254", 1, out) == G.mul(G.gen, e)
    x = [rnd.randrange(c.r) for _ in range(1 << 12)]
    d = Domain(OC["bn254"], 1 << 12)
    assert codec.fr_from_mont(cpu_oracle.ntt(0, codec.fr_to_mont(x, c), 2, threads=4), c) == d.coset_fft(x)
