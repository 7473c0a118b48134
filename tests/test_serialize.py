"""CPU: ark-serialize 0.2 layout of proofs / keys (ckb_zkp_amd/serialize.py) — round trips, decompression against the
curve equation, the y-ordering flag, infinity, Vec length prefixes and error cases.  (The byte layout itself is restated
from arkworks 0.2 — the reference holds no serialized fixture: parity unpinned, see the module docstring.)"""
import random

import pytest

from ckb_zkp_amd import serialize as ser
from ckb_zkp_amd.groth16 import Proof
from ckb_zkp_amd.params import get_curve
from oracle.pyref.curves import Group
from tests.util import OC


def _pts(curve, group, n, seed):
    G = Group(OC[curve], group)
    rnd = random.Random(seed)
    return [G.mul(G.gen, rnd.randrange(1, OC[curve].r)) for _ in range(n)]


def test_twist_constant_bn254():
    assert ser._g2_b(get_curve("bn254")) == (
        19485874751759354771024239261021720505790618469301721065564631296452457478373,
        266929791119991161246907387137283842545076965332900288569378510910307636690)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_point_round_trips_and_flags(curve):
    c = get_curve(curve)
    n = 32 if curve == "bn254" else 48
    for p in _pts(curve, 1, 12, 1) + [c.g1, None]:
        for comp in (True, False):
            b = ser.g1_to_bytes(p, curve, comp)
            assert len(b) == (n if comp else 2 * n)
            assert ser.g1_from_bytes(b, curve, comp) == p
        b = ser.g1_to_bytes(p, curve)
        if p is None:
            assert b[-1] == ser.FLAG_INFINITY and not any(b[:-1])
        else:
            assert bool(b[-1] & ser.FLAG_POSITIVE_Y) == (p[1] > c.q - p[1])
            neg = (p[0], c.q - p[1])
            bn = ser.g1_to_bytes(neg, curve)
            assert bn[:-1] == b[:-1] and (bn[-1] ^ b[-1]) == ser.FLAG_POSITIVE_Y     # same x, opposite sign flag
            assert ser.g1_from_bytes(bn, curve) == neg
    for p in _pts(curve, 2, 8, 2) + [c.g2, None]:
        for comp in (True, False):
            b = ser.g2_to_bytes(p, curve, comp)
            assert len(b) == (2 * n if comp else 4 * n)
            assert ser.g2_from_bytes(b, curve, comp) == p
        if p is not None:
            (x, (y0, y1)) = p
            neg = (x, ((c.q - y0) % c.q, (c.q - y1) % c.q))
            b, bn = ser.g2_to_bytes(p, curve), ser.g2_to_bytes(neg, curve)
            assert (bn[-1] ^ b[-1]) == ser.FLAG_POSITIVE_Y and ser.g2_from_bytes(bn, curve) == neg
            positive = (y1, y0) > ((c.q - y1) % c.q, (c.q - y0) % c.q)               # c1 is the most significant part
            assert bool(b[-1] & ser.FLAG_POSITIVE_Y) == positive


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_proof_and_parameters_round_trip(curve):
    a, cc = _pts(curve, 1, 2, 3)
    b = _pts(curve, 2, 1, 4)[0]
    pr = Proof(a, b, cc)
    raw = ser.proof_to_bytes(pr, curve)
    assert len(raw) == (128 if curve == "bn254" else 192)
    assert ser.proof_from_bytes(raw, curve) == pr
    g1, g2 = _pts(curve, 1, 20, 5), _pts(curve, 2, 8, 6)
    params = dict(vk=dict(alpha_g1=g1[0], beta_g2=g2[0], gamma_g2=g2[1], delta_g2=g2[2], gamma_abc_g1=g1[1:3]),
                  beta_g1=g1[3], delta_g1=g1[4], a_query=g1[5:9] + [None], b_g1_query=[None] + g1[9:11],
                  b_g2_query=[None] + g2[3:6], h_query=g1[11:15], l_query=g1[15:18])
    raw = ser.parameters_to_bytes(params, curve)
    assert ser.parameters_from_bytes(raw, curve) == params
    vk = ser.verify_key_to_bytes(**params["vk"], curve=curve)
    assert raw.startswith(vk) and ser.verify_key_from_bytes(vk, curve) == params["vk"]
    n = 32 if curve == "bn254" else 48
    assert raw[7 * n:7 * n + 8] == (2).to_bytes(8, "little")                       # Vec<G1Affine> length prefix


def test_rejects_bad_encodings():
    c = get_curve("bn254")
    with pytest.raises(ser.SerializationError):
        ser.g1_from_bytes(b"\x00" * 31, "bn254")
    bad = bytearray(ser.g1_to_bytes(c.g1, "bn254"))
    bad[-1] |= ser.FLAG_INFINITY | ser.FLAG_POSITIVE_Y
    with pytest.raises(ser.SerializationError):
        ser.g1_from_bytes(bytes(bad), "bn254")
    with pytest.raises(ser.SerializationError):                                     # x = 4: 4^3 + 3 = 67 is not a square mod q?
        x = next(v for v in range(2, 50) if ser._sqrt_fq(v ** 3 + 3, c.q) is None)
        ser.g1_from_bytes(x.to_bytes(32, "little"), "bn254")
    with pytest.raises(ser.SerializationError):
        ser.fr_from_bytes(c.r.to_bytes(32, "little"), "bn254")
    assert ser.fr_from_bytes(ser.fr_to_bytes(c.r - 1, "bn254"), "bn254") == c.r - 1


def test_identity_encodings():
    """compressed: x = 0 + infinity flag; uncompressed: ark's GroupAffine::zero() = (0, 1) + flag on y's last byte."""
    for curve, n in (("bn254", 32), ("bls12_381", 48)):
        b = ser.g1_to_bytes(None, curve, compressed=False)
        assert b[:n] == bytes(n) and b[n] == 1 and not any(b[n + 1:-1]) and b[-1] == ser.FLAG_INFINITY
        b2 = ser.g2_to_bytes(None, curve, compressed=False)
        assert b2[:2 * n] == bytes(2 * n) and b2[2 * n] == 1 and not any(b2[2 * n + 1:-1]) and b2[-1] == ser.FLAG_INFINITY
        assert ser.g1_from_bytes(b, curve, False) is None and ser.g2_from_bytes(b2, curve, False) is None


def _off_subgroup_point(curve, group):
    """a curve point outside the prime-order subgroup (exists where the cofactor is > 1)"""
    c = get_curve(curve)
    q = c.q
    for v in range(1, 200):
        if group == 1:
            y = ser._sqrt_fq(v ** 3 + ser._g1_b(c), q)
            p = None if y is None else (v, y)
        else:
            x = (v, 1)
            x3 = ser._fq2_mul(ser._fq2_mul(x, x, q), x, q)
            bb = ser._g2_b(c)
            y = ser._sqrt_fq2(((x3[0] + bb[0]) % q, (x3[1] + bb[1]) % q), q)
            p = None if y is None else (x, y)
        if p is not None and not ser.in_prime_order_subgroup(p, curve, group):
            return p
    raise AssertionError("no off-subgroup point found")


@pytest.mark.parametrize("curve,group", [("bn254", 2), ("bls12_381", 1), ("bls12_381", 2)])
def test_rejects_points_outside_the_prime_order_subgroup(curve, group):
    """ark-ec 0.2 `deserialize` runs is_in_correct_subgroup_assuming_on_curve: a curve point with a small-subgroup
    component (possible on BN254 G2 and both BLS12-381 groups) must not decode; deserialize_unchecked accepts it."""
    c = get_curve(curve)
    gen = c.g1 if group == 1 else c.g2
    enc, dec = (ser.g1_to_bytes, ser.g1_from_bytes) if group == 1 else (ser.g2_to_bytes, ser.g2_from_bytes)
    assert ser.in_prime_order_subgroup(gen, curve, group)
    bad = _off_subgroup_point(curve, group)
    for comp in (True, False):
        raw = enc(bad, curve, comp)
        with pytest.raises(ser.SerializationError):
            dec(raw, curve, comp)
        assert dec(raw, curve, comp, checked=False) == bad
        assert dec(enc(gen, curve, comp), curve, comp) == gen
    if group == 2:                                          # a proof whose B is off-subgroup is refused as a whole
        g1 = _pts(curve, 1, 2, 9)
        raw = ser.g1_to_bytes(g1[0], curve) + ser.g2_to_bytes(bad, curve) + ser.g1_to_bytes(g1[1], curve)
        with pytest.raises(ser.SerializationError):
            ser.proof_from_bytes(raw, curve)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_marlin_codecs_round_trip(curve):
    """zkp-marlin wire objects (marlin/src/pc/data_structures.rs:59,100,137,300; data_structures.rs:10,43): CommitterKey,
    IndexVerifierKey, Commitment (with and without the degree-bound shifted part), Proof — sizes, Option flags, round
    trips, trailing-garbage and bad-flag rejection."""
    c = get_curve(curve)
    n = 32 if curve == "bn254" else 48
    g1, g2 = _pts(curve, 1, 30, 7), _pts(curve, 2, 2, 8)
    plain, bounded = (g1[0], None), (g1[1], g1[2])
    assert len(ser.marlin_commitment_to_bytes(plain, curve)) == n + 1
    assert len(ser.marlin_commitment_to_bytes(bounded, curve)) == 2 * n + 1
    for cm in (plain, bounded, (None, None)):
        assert ser.marlin_commitment_from_bytes(ser.marlin_commitment_to_bytes(cm, curve), curve) == cm
    raw = ser.marlin_committer_key_to_bytes(g1[:5], g1[5:10], 4, curve)
    assert len(raw) == 8 + 5 * n + 8 + 5 * n + 8
    assert ser.marlin_committer_key_from_bytes(raw, curve) == dict(powers_of_g=g1[:5], powers_of_gamma_g=g1[5:10], supported_degree=4)
    ivk = dict(num_constraints=16, num_variables=16, num_non_zeros=23, index_comms=[(p, None) for p in g1[10:22]],
               g=g1[22], gamma_g=g1[23], h=g2[0], beta_h=g2[1], supported_degree=63)
    raw = ser.marlin_index_verifier_key_to_bytes(ivk, curve)
    assert len(raw) == 24 + 8 + 12 * (n + 1) + 2 * n + 4 * n + 8
    assert ser.marlin_index_verifier_key_from_bytes(raw, curve) == ivk
    rounds = [[(g1[0], None), (g1[1], None), (g1[2], None), (g1[3], None)], [(g1[4], None), (g1[5], g1[6]), (g1[7], None)],
              [(g1[8], g1[9]), (g1[10], None)]]
    evals = [5, c.r - 1, 0] + list(range(18))
    opens = [(g1[11], 12345), (g1[12], None)]
    raw = ser.marlin_proof_to_bytes(rounds, evals, opens, curve)
    assert ser.marlin_proof_from_bytes(raw, curve) == (rounds, evals, opens)
    with pytest.raises(ser.SerializationError):
        ser.marlin_proof_from_bytes(raw + b"\x00", curve)
    bad = bytearray(ser.marlin_commitment_to_bytes(plain, curve))
    bad[-1] = 2                                             # Option flag must be 0 or 1
    with pytest.raises(ser.SerializationError):
        ser.marlin_commitment_from_bytes(bytes(bad), curve)
