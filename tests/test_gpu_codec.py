"""ark-serialize point codec behind the C ABI (zkp_g1/g2_compress / _decompress / _subgroup_check, csrc/msm_group.hip "point
codec") and the product's container framing (ckb_zkp_amd/serialize.py *_abi) against the ORACLE's restatement of the layout
(oracle/pyref/serialize.py, written independently of the product's host codec; both restate ark-serialize 0.2 — the reference
holds no serialized fixture, so byte parity with a real .pk stays unpinned), and through a full Parameters round trip:
key -> `Parameters::serialize` bytes -> device decompression -> proving key -> the same proof."""
import random

import numpy as np
import pytest

from ckb_zkp_amd import codec, groth16, serialize
from ckb_zkp_amd.circuits import mimc_chain_instance
from ckb_zkp_amd.params import get_curve
from oracle.pyref import serialize as oser
from tests.util import OC

TOXIC = dict(alpha=0x1234567890ABCDEF1, beta=0xFEDCBA09876543211, gamma=0x1111111111111111111, delta=0x2222222222222222223, tau=0x3333333333333333335)

pytestmark = pytest.mark.gpu


def _points(curve, group, n, seed):
    from oracle.pyref.curves import Group
    G = Group(OC[curve], group)
    rnd = random.Random(seed)
    pts = [G.mul(G.gen, rnd.randrange(1, G.order)) for _ in range(n)]
    pts[n // 2] = None                                           # the identity
    return pts


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
@pytest.mark.parametrize("group", [1, 2])
def test_compress_decompress_match_the_oracle_codec(ctx, curve, group):
    c = get_curve(curve)
    pts = _points(curve, group, 40, 11 * group + len(curve))
    to_b = lambda p, _c: oser.point_encode(p, OC[curve], group)
    to_m = codec.g1_to_mont if group == 1 else codec.g2_to_mont
    want = b"".join(to_b(p, c) for p in pts)
    xy, inf = to_m(pts, c)
    assert ctx.compress_points(c, group, xy, inf) == want
    xy2, inf2 = ctx.decompress_points(c, group, want)
    assert np.array_equal(xy2, xy) and list(inf2) == list(inf)
    # also the negated points (the other y flag)
    from oracle.pyref.curves import Group
    G = Group(OC[curve], group)
    neg = [G.neg(p) for p in pts]
    want_n = b"".join(to_b(p, c) for p in neg)
    xy_n, inf_n = to_m(neg, c)
    assert ctx.compress_points(c, group, xy_n, inf_n) == want_n
    assert np.array_equal(ctx.decompress_points(c, group, want_n)[0], xy_n)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_malformed_points_are_rejected_with_their_index(ctx, curve):
    c = get_curve(curve)
    n = 8 * c.fq_limbs
    good = oser.point_encode(_points(curve, 1, 3, 5)[0], OC[curve], 1)
    # (a) both flags set, (b) x >= p, (c) x^3 + b is not a square, (d) x >= p WITH the infinity flag (ark range-checks the field
    # element before it looks at the flags: ADVICE r3) — each one is refused by the oracle codec too
    both = bytearray(good)
    both[-1] |= 0xC0
    big = bytearray((c.q + 1).to_bytes(n, "little"))
    x = 1
    while pow((x * x * x + (3 if curve == "bn254" else 4)) % c.q, (c.q - 1) // 2, c.q) == 1:
        x += 1
    nonres = x.to_bytes(n, "little")
    big_inf = bytearray(big)
    big_inf[-1] |= oser.INFINITY
    for k, bad in enumerate((bytes(both), bytes(big), nonres, bytes(big_inf))):
        with pytest.raises(oser.InvalidData):
            oser.point_decode(bad, OC[curve], 1)
        data = good * (k + 1) + bad + good
        with pytest.raises(ValueError) as e:
            ctx.decompress_points(c, 1, data)
        assert f"index {k + 1}" in str(e.value)
    # G2: a non-canonical c0 under the infinity flag (which lives on c1's last byte)
    enc = bytearray(c.q.to_bytes(n, "little") + bytes(n))
    enc[-1] |= oser.INFINITY
    g2_good = oser.point_encode(_points(curve, 2, 3, 6)[0], OC[curve], 2)
    with pytest.raises(oser.InvalidData):
        oser.point_decode(bytes(enc), OC[curve], 2)
    with pytest.raises(ValueError) as e:
        ctx.decompress_points(c, 2, g2_good + bytes(enc))
    assert "index 1" in str(e.value)
    # the canonical identity encodings pass
    xy, inf = ctx.decompress_points(c, 2, g2_good + oser.point_encode(None, OC[curve], 2))
    assert list(inf) == [0, 1] and not xy[1].any()


def test_argument_errors_are_not_reported_as_point_zero(ctx):
    """ZKP_ERR_INVALID_POINT (with an index) is distinct from ZKP_ERR_BAD_ARG (index = SIZE_MAX), and a failed
    decompression leaves the output arrays untouched"""
    import ctypes as C
    from ckb_zkp_amd import _lib
    c = get_curve("bn254")
    bad = C.c_size_t(0)
    xy = np.full((2, 8), 7, dtype=np.uint64)
    inf = np.full(2, 9, dtype=np.uint8)
    rc = ctx.lib.zkp_g1_decompress(ctx.h, c.cid, None, 2, C.c_void_p(xy.ctypes.data), C.c_void_p(inf.ctypes.data), C.byref(bad))
    assert rc == -1 and bad.value == C.c_size_t(-1).value
    data = np.frombuffer(oser.point_encode(_points("bn254", 1, 3, 5)[0], OC["bn254"], 1) + b"\xff" * 32, dtype=np.uint8)
    rc = ctx.lib.zkp_g1_decompress(ctx.h, c.cid, C.c_void_p(data.ctypes.data), 2, C.c_void_p(xy.ctypes.data),
                                   C.c_void_p(inf.ctypes.data), C.byref(bad))
    assert rc == _lib.ZKP_ERR_INVALID_POINT and bad.value == 1
    assert (xy == 7).all() and (inf == 9).all()
    assert b"InvalidData" in ctx.lib.zkp_status_string(rc)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_parameters_bytes_roundtrip_proves_the_same(ctx, curve):
    c = get_curve(curve)
    inst = mimc_chain_instance(curve, 40)
    params = groth16.generate_parameters(ctx, curve, inst, **TOXIC)
    blob = serialize.parameters_to_bytes_abi(ctx, params)
    # the oracle codec reads the same bytes to the same points, and writes the same bytes back
    ref = oser.parameters_decode(blob, OC[curve], checked=False)
    assert ref["a_query"] == codec.g1_from_mont(params.a_query[0], params.a_query[1], c)
    assert ref["b_g2_query"] == codec.g2_from_mont(params.b_g2_query[0], params.b_g2_query[1], c)
    assert ref["vk"]["gamma_abc_g1"] == codec.g1_from_mont(params.gamma_abc_g1[0], params.gamma_abc_g1[1], c)
    assert oser.parameters_encode(ref, OC[curve]) == blob
    loaded = serialize.parameters_from_bytes_abi(ctx, blob, curve, inst.num_constraints())
    for name in ("a_query", "b_g1_query", "b_g2_query", "h_query", "l_query", "gamma_abc_g1"):
        a, b = getattr(params, name), getattr(loaded, name)
        assert np.array_equal(a[0], b[0]) and list(a[1]) == list(b[1]), name
    for name in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "gamma_g2", "delta_g2"):
        assert np.array_equal(getattr(params, name), getattr(loaded, name)), name
    z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
    r, s = codec.fr_to_mont([0x1234567], c)[0], codec.fr_to_mont([0x7654321], c)[0]
    pk1, pk2 = groth16.ProvingKey(ctx, params, inst), groth16.ProvingKey(ctx, loaded, inst)
    p1, i1 = pk1.prove_raw(z, r, s)
    p2, i2 = pk2.prove_raw(z, r, s)
    assert np.array_equal(p1, p2) and list(i1) == list(i2)
    # Proof::serialize: ABI path == Python restatement of the same proof
    w = 2 * c.fq_limbs
    pa = codec.g1_from_mont(p1[:w].reshape(1, -1), [i1[0]], c)[0]
    pb = codec.g2_from_mont(p1[w:3 * w].reshape(1, -1), [i1[1]], c)[0]
    pc = codec.g1_from_mont(p1[3 * w:].reshape(1, -1), [i1[2]], c)[0]
    assert serialize.proof_to_bytes_abi(ctx, curve, p1, i1) == oser.proof_encode(pa, pb, pc, OC[curve])
    pk1.free()
    pk2.free()


# ------------------------------------------------------------------------------------------- checked deserialize (subgroup)
def _curve_point_outside_subgroup(curve, group):
    """a point ON the curve whose order does not divide r (cofactor groups only): decompress small x values until one works"""
    from oracle.pyref.curves import Group
    c = get_curve(curve)
    n = 8 * c.fq_limbs
    G = Group(OC[curve], group)
    x = 1
    while True:
        data = x.to_bytes(n, "little") + (b"" if group == 1 else (1).to_bytes(n, "little"))
        x += 1
        try:
            P = oser.point_decode(data, OC[curve], group, checked=False)
        except oser.InvalidData:
            continue
        if P is not None and G.on_curve(P) and not oser.in_subgroup(P, OC[curve], group):
            return P


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
@pytest.mark.parametrize("group", [1, 2])
def test_subgroup_check_accepts_subgroup_points_and_rejects_off_curve(ctx, curve, group):
    c = get_curve(curve)
    pts = _points(curve, group, 24, 3 * group + len(curve))
    to_m = codec.g1_to_mont if group == 1 else codec.g2_to_mont
    xy, inf = to_m(pts, c)
    ctx.subgroup_check(c, group, xy, inf)                         # incl. the identity at index 12
    ctx.subgroup_check(c, group, xy[:0], inf[:0])                 # empty
    bad = xy.copy()
    bad[7, -1] ^= 1                                               # y (G2: its c1) off by one Montgomery unit: not on the curve
    with pytest.raises(ValueError) as e:
        ctx.subgroup_check(c, group, bad, inf)
    assert "point 7 " in str(e.value)


@pytest.mark.parametrize("curve,group", [("bls12_381", 1), ("bls12_381", 2), ("bn254", 2)])
def test_subgroup_check_rejects_curve_points_outside_the_subgroup(ctx, curve, group):
    c = get_curve(curve)
    P = _curve_point_outside_subgroup(curve, group)
    pts = _points(curve, group, 12, 17)
    pts[9] = P
    pts[10] = P                                                   # the FIRST failing index is reported
    to_m = codec.g1_to_mont if group == 1 else codec.g2_to_mont
    xy, inf = to_m(pts, c)
    with pytest.raises(ValueError) as e:
        ctx.subgroup_check(c, group, xy, inf)
    assert "point 9 " in str(e.value)
    # the same verdict as the oracle's restatement of ark's check
    assert not oser.in_subgroup(P, OC[curve], group)
    assert oser.in_subgroup(pts[0], OC[curve], group)


def test_checked_parameters_load_rejects_a_small_subgroup_element(ctx):
    curve = "bls12_381"
    c = get_curve(curve)
    inst = mimc_chain_instance(curve, 12)
    params = groth16.generate_parameters(ctx, curve, inst, **TOXIC)
    blob = serialize.parameters_to_bytes_abi(ctx, params)
    serialize.parameters_from_bytes_abi(ctx, blob, curve, inst.num_constraints())                 # checked: passes
    ref = oser.parameters_decode(blob, OC[curve], checked=False)
    ref["b_g2_query"][3] = _curve_point_outside_subgroup(curve, 2)
    blob2 = oser.parameters_encode(ref, OC[curve])
    with pytest.raises(oser.InvalidData):
        oser.parameters_decode(blob2, OC[curve], checked=True)
    with pytest.raises(serialize.SerializationError):
        serialize.parameters_from_bytes_abi(ctx, blob2, curve, inst.num_constraints())
    loaded = serialize.parameters_from_bytes_abi(ctx, blob2, curve, inst.num_constraints(), checked=False)
    assert loaded.b_g2_query[0].shape == params.b_g2_query[0].shape
