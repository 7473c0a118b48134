"""ark-serialize 0.2 wire format for the objects that cross the prover boundary on disk / on the wire
(SURVEY.md §8(f)-1): `Proof`, `VerifyKey`, `Parameters` of zkp-groth16 (derives at /root/reference/groth16/src/lib.rs:51,
59,81; written by cli/src/setup.rs:41-45,76-83 and cli/src/zkp_prove.rs:45-49,116-124) and the KZG10 commitments of
zkp-marlin.

Layout (arkworks `CanonicalSerialize`, compressed form — the default `serialize`):
  * prime-field element: the canonical (non-Montgomery) integer, little-endian, ceil(bits/8) bytes (32 for both Fr and
    BN254 Fq, 48 for BLS12-381 Fq);
  * Fq2: c0 then c1;
  * short-Weierstrass affine point: x only, with two flag bits in the top of the LAST byte: bit 7 = "y is the larger of
    {y, -y}" (Fq: as integers; Fq2: compare c1 first, then c0), bit 6 = point at infinity (x = 0);
  * uncompressed form: x, then y, the infinity flag on y's last byte; the identity is written with ark's in-memory
    coordinates of `GroupAffine::zero()` = (x = 0, y = 1) plus the flag (compressed: x = 0 plus the flag);
  * `deserialize` (compressed and uncompressed) rejects points outside the prime-order subgroup
    (`is_in_correct_subgroup_assuming_on_curve`): relevant for BN254 G2 and both BLS12-381 groups (BN254 G1 has
    cofactor 1).  `checked=False` is ark's `deserialize_unchecked` (bulk loads of trusted keys);
  * `Vec<T>`: u64 little-endian length, then the elements; structs: their fields in declaration order.

PARITY UNPINNED: the reference holds no serialized fixture and ark-serialize is not vendored (Cargo dependency "0.2"),
so this layout is restated from the published arkworks 0.2 sources and checked here by round trips, by the curve
equation on decompression and by the y-ordering rule; confirm against real `.pk` / proof bytes when a Rust toolchain
exists.
"""
from __future__ import annotations

import struct

from .params import CurveParams, get_curve

FLAG_POSITIVE_Y = 1 << 7
FLAG_INFINITY = 1 << 6


class SerializationError(Exception):
    """ark_serialize::SerializationError (InvalidData / UnexpectedFlags / NotEnoughSpace)"""


# ------------------------------------------------------------------ field helpers (host-side big integers)
def _fq_bytes(c: CurveParams) -> int:
    return (c.q.bit_length() + 2 + 7) // 8          # modulus bits + 2 flag bits


def _fr_bytes(c: CurveParams) -> int:
    return (c.r.bit_length() + 7) // 8


def _sqrt_fq(a: int, q: int):
    """q = 3 mod 4 for both base fields"""
    a %= q
    if a == 0:
        return 0
    s = pow(a, (q + 1) // 4, q)
    return s if s * s % q == a else None


def _fq2_mul(a, b, q):
    return ((a[0] * b[0] - a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)


def _sqrt_fq2(a, q):
    """square root in Fq[u]/(u^2 + 1) by the norm method"""
    a0, a1 = a[0] % q, a[1] % q
    if a1 == 0:
        s = _sqrt_fq(a0, q)
        if s is not None:
            return (s, 0)
        s = _sqrt_fq(-a0, q)                         # a0 = -(s^2) = (s u)^2
        return None if s is None else (0, s)
    n = _sqrt_fq(a0 * a0 + a1 * a1, q)               # norm
    if n is None:
        return None
    inv2 = pow(2, -1, q)
    for sign in (1, -1):
        x2 = (a0 + sign * n) * inv2 % q
        x = _sqrt_fq(x2, q)
        if x is not None and x != 0:
            y = a1 * pow(2 * x, -1, q) % q
            if _fq2_mul((x, y), (x, y), q) == (a0, a1):
                return (x, y)
    return None


def _g2_b(c: CurveParams):
    """twist coefficient: BN254 3/(9+u), BLS12-381 4(1+u)"""
    q = c.q
    if c.name == "bn254":
        d = pow(9 * 9 + 1, -1, q)                    # 1/(9+u) = (9-u)/82
        return (3 * 9 * d % q, (-3 * d) % q)
    return (4, 4)


def _g1_b(c: CurveParams) -> int:
    return 3 if c.name == "bn254" else 4


# ---- prime-order subgroup membership: [r]P == O, affine double-and-add on host big integers (a = 0 curves)
def _fq2_inv(a, q):
    d = pow((a[0] * a[0] + a[1] * a[1]) % q, -1, q)
    return (a[0] * d % q, (-a[1] * d) % q)


class _F1:
    def __init__(self, q): self.q = q
    def add(self, a, b): return (a + b) % self.q
    def sub(self, a, b): return (a - b) % self.q
    def mul(self, a, b): return a * b % self.q
    def inv(self, a): return pow(a, -1, self.q)
    def small(self, k, a): return k * a % self.q
    zero = 0


class _F2:
    def __init__(self, q): self.q = q
    def add(self, a, b): return ((a[0] + b[0]) % self.q, (a[1] + b[1]) % self.q)
    def sub(self, a, b): return ((a[0] - b[0]) % self.q, (a[1] - b[1]) % self.q)
    def mul(self, a, b): return _fq2_mul(a, b, self.q)
    def inv(self, a): return _fq2_inv(a, self.q)
    def small(self, k, a): return (k * a[0] % self.q, k * a[1] % self.q)
    zero = (0, 0)


def _affine_add(F, p, q_):
    if p is None:
        return q_
    if q_ is None:
        return p
    (x1, y1), (x2, y2) = p, q_
    if x1 == x2:
        if F.add(y1, y2) == F.zero:
            return None
        lam = F.mul(F.small(3, F.mul(x1, x1)), F.inv(F.small(2, y1)))
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    return (x3, F.sub(F.mul(lam, F.sub(x1, x3)), y1))


def in_prime_order_subgroup(p, curve, group: int) -> bool:
    """ark-ec `is_in_correct_subgroup_assuming_on_curve`: [r]P is the identity."""
    c = get_curve(curve)
    if p is None or (group == 1 and c.name == "bn254"):          # cofactor 1
        return True
    F = _F1(c.q) if group == 1 else _F2(c.q)
    acc = None
    for bit in bin(c.r)[2:]:
        acc = _affine_add(F, acc, acc)
        if bit == "1":
            acc = _affine_add(F, acc, p)
    return acc is None


# ------------------------------------------------------------------ elements
def fr_to_bytes(x: int, curve) -> bytes:
    c = get_curve(curve)
    return (x % c.r).to_bytes(_fr_bytes(c), "little")


def fr_from_bytes(b: bytes, curve) -> int:
    c = get_curve(curve)
    if len(b) != _fr_bytes(c):
        raise SerializationError("NotEnoughSpace")
    x = int.from_bytes(b, "little")
    if x >= c.r:
        raise SerializationError("InvalidData")
    return x


def _y_is_positive_g1(y: int, q: int) -> bool:
    return y > (q - y) % q


def _y_is_positive_g2(y, q: int) -> bool:
    ny = ((q - y[0]) % q, (q - y[1]) % q)
    return (y[1], y[0]) > (ny[1], ny[0])             # QuadExtField Ord: c1 first, then c0


def g1_to_bytes(p, curve, compressed: bool = True) -> bytes:
    c = get_curve(curve)
    n = _fq_bytes(c)
    if p is None:
        out = bytearray(n if compressed else 2 * n)
        if not compressed:
            out[n] = 1                                # GroupAffine::zero() = (0, 1, infinity)
        out[-1] |= FLAG_INFINITY
        return bytes(out)
    x, y = p
    if compressed:
        out = bytearray(x.to_bytes(n, "little"))
        if _y_is_positive_g1(y, c.q):
            out[-1] |= FLAG_POSITIVE_Y
        return bytes(out)
    return x.to_bytes(n, "little") + y.to_bytes(n, "little")


def g1_from_bytes(b: bytes, curve, compressed: bool = True, checked: bool = True):
    c = get_curve(curve)
    n, q = _fq_bytes(c), c.q
    if len(b) != (n if compressed else 2 * n):
        raise SerializationError("NotEnoughSpace")
    flags = b[-1] & (FLAG_POSITIVE_Y | FLAG_INFINITY)
    if flags == (FLAG_POSITIVE_Y | FLAG_INFINITY):
        raise SerializationError("UnexpectedFlags")
    body = bytearray(b)
    body[-1] &= 0x3F
    # ark reads the field element(s) first (deserialize_with_flags -> Fp::read -> from_repr): a non-canonical coordinate is
    # InvalidData even when the infinity flag is set
    if any(int.from_bytes(body[i:i + n], "little") >= q for i in range(0, len(body), n)):
        raise SerializationError("InvalidData")
    if flags & FLAG_INFINITY:
        return None
    if compressed:
        x = int.from_bytes(body, "little")
        y = _sqrt_fq(x * x * x + _g1_b(c), q)
        if y is None:
            raise SerializationError("InvalidData")
        if _y_is_positive_g1(y, q) != bool(flags & FLAG_POSITIVE_Y):
            y = (q - y) % q
    else:
        x, y = int.from_bytes(body[:n], "little"), int.from_bytes(body[n:], "little")
        if x >= q or y >= q or (y * y - x * x * x - _g1_b(c)) % q:
            raise SerializationError("InvalidData")
    if checked and not in_prime_order_subgroup((x, y), c, 1):
        raise SerializationError("InvalidData")
    return (x, y)


def g2_to_bytes(p, curve, compressed: bool = True) -> bytes:
    c = get_curve(curve)
    n = _fq_bytes(c)
    if p is None:
        out = bytearray(2 * n if compressed else 4 * n)
        if not compressed:
            out[2 * n] = 1                            # y = Fq2::one() = (1, 0)
        out[-1] |= FLAG_INFINITY
        return bytes(out)
    (x0, x1), (y0, y1) = p
    if compressed:
        out = bytearray(x0.to_bytes(n, "little") + x1.to_bytes(n, "little"))
        if _y_is_positive_g2((y0, y1), c.q):
            out[-1] |= FLAG_POSITIVE_Y
        return bytes(out)
    return b"".join(v.to_bytes(n, "little") for v in (x0, x1, y0, y1))


def g2_from_bytes(b: bytes, curve, compressed: bool = True, checked: bool = True):
    c = get_curve(curve)
    n, q = _fq_bytes(c), c.q
    if len(b) != (2 * n if compressed else 4 * n):
        raise SerializationError("NotEnoughSpace")
    flags = b[-1] & (FLAG_POSITIVE_Y | FLAG_INFINITY)
    if flags == (FLAG_POSITIVE_Y | FLAG_INFINITY):
        raise SerializationError("UnexpectedFlags")
    body = bytearray(b)
    body[-1] &= 0x3F
    vals = [int.from_bytes(body[i * n:(i + 1) * n], "little") for i in range(len(body) // n)]
    if any(v >= q for v in vals):                    # before the infinity flag, like ark (see g1_from_bytes)
        raise SerializationError("InvalidData")
    if flags & FLAG_INFINITY:
        return None
    x = (vals[0], vals[1])
    x3 = _fq2_mul(_fq2_mul(x, x, q), x, q)
    bb = _g2_b(c)
    rhs = ((x3[0] + bb[0]) % q, (x3[1] + bb[1]) % q)
    if compressed:
        y = _sqrt_fq2(rhs, q)
        if y is None:
            raise SerializationError("InvalidData")
        if _y_is_positive_g2(y, q) != bool(flags & FLAG_POSITIVE_Y):
            y = ((q - y[0]) % q, (q - y[1]) % q)
    else:
        y = (vals[2], vals[3])
        if _fq2_mul(y, y, q) != rhs:
            raise SerializationError("InvalidData")
    if checked and not in_prime_order_subgroup((x, y), c, 2):
        raise SerializationError("InvalidData")
    return (x, y)


# ------------------------------------------------------------------ containers
class _Reader:
    def __init__(self, b: bytes):
        self.b, self.o = memoryview(b), 0

    def take(self, n: int) -> bytes:
        if self.o + n > len(self.b):
            raise SerializationError("NotEnoughSpace")
        out = bytes(self.b[self.o:self.o + n])
        self.o += n
        return out

    def u64(self) -> int:
        return struct.unpack("<Q", self.take(8))[0]


def _vec(items, enc) -> bytes:
    return struct.pack("<Q", len(items)) + b"".join(enc(x) for x in items)


def proof_to_bytes(proof, curve) -> bytes:
    """Proof { a: G1Affine, b: G2Affine, c: G1Affine }  (groth16/src/lib.rs:51-56): 128 B on BN254, 192 B on BLS12-381"""
    return g1_to_bytes(proof.a, curve) + g2_to_bytes(proof.b, curve) + g1_to_bytes(proof.c, curve)


def proof_from_bytes(b: bytes, curve):
    """`Proof::deserialize` (checked: an untrusted proof with a small-subgroup component is rejected)"""
    from .groth16 import Proof
    c = get_curve(curve)
    n = _fq_bytes(c)
    if len(b) != 4 * n:
        raise SerializationError("NotEnoughSpace")
    return Proof(g1_from_bytes(b[:n], c), g2_from_bytes(b[n:3 * n], c), g1_from_bytes(b[3 * n:], c))


def verify_key_to_bytes(alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1, curve) -> bytes:
    """VerifyKey { alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1: Vec<G1Affine> }  (lib.rs:59-66)"""
    return (g1_to_bytes(alpha_g1, curve) + g2_to_bytes(beta_g2, curve) + g2_to_bytes(gamma_g2, curve) +
            g2_to_bytes(delta_g2, curve) + _vec(gamma_abc_g1, lambda p: g1_to_bytes(p, curve)))


def _read_verify_key(r: _Reader, c: CurveParams, checked: bool = True) -> dict:
    n = _fq_bytes(c)
    k = dict(checked=checked)
    vk = dict(alpha_g1=g1_from_bytes(r.take(n), c, **k), beta_g2=g2_from_bytes(r.take(2 * n), c, **k),
              gamma_g2=g2_from_bytes(r.take(2 * n), c, **k), delta_g2=g2_from_bytes(r.take(2 * n), c, **k))
    vk["gamma_abc_g1"] = [g1_from_bytes(r.take(n), c, **k) for _ in range(r.u64())]
    return vk


def verify_key_from_bytes(b: bytes, curve, checked: bool = True) -> dict:
    r = _Reader(b)
    vk = _read_verify_key(r, get_curve(curve), checked)
    if r.o != len(b):
        raise SerializationError("InvalidData")
    return vk


def parameters_to_bytes(p: dict, curve) -> bytes:
    """Parameters { vk, beta_g1, delta_g1, a_query, b_g1_query, b_g2_query, h_query, l_query }  (lib.rs:81-91).
    p: dict of canonical points (None = infinity) with the reference's field names; `vk` a dict as above."""
    g1 = lambda q: g1_to_bytes(q, curve)
    g2 = lambda q: g2_to_bytes(q, curve)
    vk = p["vk"]
    return (verify_key_to_bytes(vk["alpha_g1"], vk["beta_g2"], vk["gamma_g2"], vk["delta_g2"], vk["gamma_abc_g1"], curve) +
            g1(p["beta_g1"]) + g1(p["delta_g1"]) + _vec(p["a_query"], g1) + _vec(p["b_g1_query"], g1) +
            _vec(p["b_g2_query"], g2) + _vec(p["h_query"], g1) + _vec(p["l_query"], g1))


def parameters_from_bytes(b: bytes, curve, checked: bool = True) -> dict:
    """`Parameters::deserialize`; checked=False = `deserialize_unchecked`.  This is the pure-Python restatement (a big-integer
    square root and, when checked, a 255-step affine double-and-add PER POINT): the reference for small keys and tests.  A real key
    goes through `parameters_from_bytes_abi`, which decompresses and subgroup-checks on the device (zkp_g*_decompress,
    zkp_g*_subgroup_check): 2^20-point queries in well under a second, checked."""
    c = get_curve(curve)
    n = _fq_bytes(c)
    r = _Reader(b)
    out = dict(vk=_read_verify_key(r, c, checked))
    out["beta_g1"] = g1_from_bytes(r.take(n), c, checked=checked)
    out["delta_g1"] = g1_from_bytes(r.take(n), c, checked=checked)
    for name, size, dec in (("a_query", n, g1_from_bytes), ("b_g1_query", n, g1_from_bytes),
                            ("b_g2_query", 2 * n, g2_from_bytes), ("h_query", n, g1_from_bytes),
                            ("l_query", n, g1_from_bytes)):
        out[name] = [dec(r.take(size), c, checked=checked) for _ in range(r.u64())]
    if r.o != len(b):
        raise SerializationError("InvalidData")
    return out


# ------------------------------------------------------------------ zkp-marlin (CanonicalSerialize derives)
# marlin/src/pc/data_structures.rs:59-65 CommitterKey, :100-109 VerifierKey, :121-122 Comm, :137-141 Commitment,
# :300-304 pc::Proof; marlin/src/ahp/indexer.rs:12-17 IndexInfo; marlin/src/data_structures.rs:10-15 IndexVerifierKey,
# :43-48 Proof.  Derived impls write the fields in declaration order; `usize` as u64; `Option<T>` as a bool byte followed
# by the value when present; `bool` as one byte (ark-serialize 0.2, restated — PARITY UNPINNED like the rest of this file).
def _u64(v: int) -> bytes:
    return struct.pack("<Q", v)


def _option(v, enc) -> bytes:
    return b"\x00" if v is None else b"\x01" + enc(v)


def _read_option(r: _Reader, dec):
    flag = r.take(1)[0]
    if flag > 1:
        raise SerializationError("InvalidData")
    return dec() if flag else None


def marlin_commitment_to_bytes(comm, curve) -> bytes:
    """pc::Commitment { comm: Comm(G1Affine), shifted_comm: Option<Comm> }; comm = (point, shifted point or None)"""
    c, shifted = comm
    return g1_to_bytes(c, curve) + _option(shifted, lambda p: g1_to_bytes(p, curve))


def _read_marlin_commitment(r: _Reader, c: CurveParams, checked: bool):
    n = _fq_bytes(c)
    cm = g1_from_bytes(r.take(n), c, checked=checked)
    return cm, _read_option(r, lambda: ("some", g1_from_bytes(r.take(n), c, checked=checked)))


def marlin_commitment_from_bytes(b: bytes, curve, checked: bool = True):
    r = _Reader(b)
    cm, sh = _read_marlin_commitment(r, get_curve(curve), checked)
    if r.o != len(b):
        raise SerializationError("InvalidData")
    return cm, (sh[1] if sh else None)


def marlin_committer_key_to_bytes(powers_of_g, powers_of_gamma_g, supported_degree: int, curve) -> bytes:
    g1 = lambda p: g1_to_bytes(p, curve)
    return _vec(powers_of_g, g1) + _vec(powers_of_gamma_g, g1) + _u64(supported_degree)


def marlin_committer_key_from_bytes(b: bytes, curve, checked: bool = True) -> dict:
    c = get_curve(curve)
    n = _fq_bytes(c)
    r = _Reader(b)
    out = dict(powers_of_g=[g1_from_bytes(r.take(n), c, checked=checked) for _ in range(r.u64())])
    out["powers_of_gamma_g"] = [g1_from_bytes(r.take(n), c, checked=checked) for _ in range(r.u64())]
    out["supported_degree"] = r.u64()
    if r.o != len(b):
        raise SerializationError("InvalidData")
    return out


def marlin_verifier_key_to_bytes(vk: dict, curve) -> bytes:
    """pc::VerifierKey { g, gamma_g, h, beta_h, supported_degree }"""
    return (g1_to_bytes(vk["g"], curve) + g1_to_bytes(vk["gamma_g"], curve) + g2_to_bytes(vk["h"], curve) +
            g2_to_bytes(vk["beta_h"], curve) + _u64(vk["supported_degree"]))


def marlin_index_verifier_key_to_bytes(ivk: dict, curve) -> bytes:
    """IndexVerifierKey { index_info { num_constraints, num_variables, num_non_zeros }, index_comms: Vec<Commitment>,
    verifier_key }"""
    return (_u64(ivk["num_constraints"]) + _u64(ivk["num_variables"]) + _u64(ivk["num_non_zeros"]) +
            _vec(ivk["index_comms"], lambda cm: marlin_commitment_to_bytes(cm, curve)) + marlin_verifier_key_to_bytes(ivk, curve))


def marlin_index_verifier_key_from_bytes(b: bytes, curve, checked: bool = True) -> dict:
    c = get_curve(curve)
    n = _fq_bytes(c)
    r = _Reader(b)
    out = dict(num_constraints=r.u64(), num_variables=r.u64(), num_non_zeros=r.u64())
    comms = []
    for _ in range(r.u64()):
        cm, sh = _read_marlin_commitment(r, c, checked)
        comms.append((cm, sh[1] if sh else None))
    out["index_comms"] = comms
    out["g"] = g1_from_bytes(r.take(n), c, checked=checked)
    out["gamma_g"] = g1_from_bytes(r.take(n), c, checked=checked)
    out["h"] = g2_from_bytes(r.take(2 * n), c, checked=checked)
    out["beta_h"] = g2_from_bytes(r.take(2 * n), c, checked=checked)
    out["supported_degree"] = r.u64()
    if r.o != len(b):
        raise SerializationError("InvalidData")
    return out


def marlin_proof_to_bytes(commitments, evaluations, opening_proofs, curve) -> bytes:
    """marlin::Proof { commitments: Vec<Vec<Commitment>>, evaluations: Vec<Fr>, opening_proofs: Vec<pc::Proof> } with
    pc::Proof { w: G1Affine, rand_v: Option<Fr> }.  commitments: the three rounds' lists of (point, shifted or None)."""
    cm = lambda x: marlin_commitment_to_bytes(x, curve)
    out = _vec(commitments, lambda rnd: _vec(rnd, cm))
    out += _vec(evaluations, lambda e: fr_to_bytes(e, curve))
    out += _vec(opening_proofs, lambda p: g1_to_bytes(p[0], curve) + _option(p[1], lambda v: fr_to_bytes(v, curve)))
    return out


def marlin_proof_from_bytes(b: bytes, curve, checked: bool = True):
    c = get_curve(curve)
    n, fr = _fq_bytes(c), _fr_bytes(c)
    r = _Reader(b)
    commitments = []
    for _ in range(r.u64()):
        rnd = []
        for _ in range(r.u64()):
            cm, sh = _read_marlin_commitment(r, c, checked)
            rnd.append((cm, sh[1] if sh else None))
        commitments.append(rnd)
    evaluations = [fr_from_bytes(r.take(fr), c) for _ in range(r.u64())]
    proofs = []
    for _ in range(r.u64()):
        w = g1_from_bytes(r.take(n), c, checked=checked)
        proofs.append((w, _read_option(r, lambda: fr_from_bytes(r.take(fr), c))))
    if r.o != len(b):
        raise SerializationError("InvalidData")
    return commitments, evaluations, proofs


# ------------------------------------------------------------------ bulk codecs over the C ABI (device decompression)
# `Parameters::serialize` / `deserialize_unchecked` of a multi-million-point key without a Python big-integer per point: the
# container framing is parsed here, the point vectors go through zkp_g1/g2_compress / zkp_g1/g2_decompress (one lane per point
# on the GPU: a square root each).  Same bytes as parameters_to_bytes / parameters_from_bytes above (tests/test_gpu_codec.py).
def parameters_to_bytes_abi(ctx, params) -> bytes:
    """groth16.Parameters (ABI arrays) -> the bytes of `Parameters::serialize` (cli/src/setup.rs:41-45)"""
    import numpy as np
    c = params.curve
    one1 = lambda xy: ctx.compress_points(c, 1, np.ascontiguousarray(xy).reshape(1, -1))
    one2 = lambda xy: ctx.compress_points(c, 2, np.ascontiguousarray(xy).reshape(1, -1))
    vec = lambda g, q: struct.pack("<Q", q[0].shape[0]) + ctx.compress_points(c, g, q[0], q[1])
    return (one1(params.alpha_g1) + one2(params.beta_g2) + one2(params.gamma_g2) + one2(params.delta_g2) + vec(1, params.gamma_abc_g1) +
            one1(params.beta_g1) + one1(params.delta_g1) + vec(1, params.a_query) + vec(1, params.b_g1_query) +
            vec(2, params.b_g2_query) + vec(1, params.h_query) + vec(1, params.l_query))


def parameters_from_bytes_abi(ctx, b: bytes, curve, num_constraints: int, checked: bool = True):
    """`Parameters::deserialize` (checked=True: every point on the curve and in the prime-order subgroup, like ark-ec 0.2's
    `GroupAffine::deserialize`; False = `deserialize_unchecked`) -> groth16.Parameters (ABI arrays).  Decompression and the subgroup
    check run on the device (zkp_g*_decompress / zkp_g*_subgroup_check): a 2^20 key loads checked in well under a second where the
    pure-Python `parameters_from_bytes` needs hours.
    num_constraints is not part of the file (the reference re-synthesises the circuit): the caller supplies it."""
    from .groth16 import Parameters
    c = get_curve(curve)
    n = _fq_bytes(c)
    r = _Reader(b)

    def pts(group, count):
        try:
            xy, inf = ctx.decompress_points(c, group, r.take(count * n * group))
            if checked and not (group == 1 and c.name == "bn254"):       # BN254 G1 has cofactor 1
                ctx.subgroup_check(c, group, xy, inf)
            return xy, inf
        except ValueError as e:
            raise SerializationError(f"InvalidData ({e})")

    def single(group):
        # ark's `Parameters::deserialize` accepts the identity for alpha / beta / gamma / delta (a useless key, not a malformed
        # one); in the ABI layout the identity is the all-zero coordinate pair, which the prover's tables treat as such
        xy, _ = pts(group, 1)
        return xy[0]

    alpha_g1, beta_g2, gamma_g2, delta_g2 = single(1), single(2), single(2), single(2)
    gamma_abc = pts(1, r.u64())
    beta_g1, delta_g1 = single(1), single(1)
    a_q = pts(1, r.u64())
    b1_q = pts(1, r.u64())
    b2_q = pts(2, r.u64())
    h_q = pts(1, r.u64())
    l_q = pts(1, r.u64())
    if r.o != len(b):
        raise SerializationError("InvalidData")
    return Parameters(curve=c, num_inputs=gamma_abc[0].shape[0], num_aux=l_q[0].shape[0], num_constraints=num_constraints,
                      alpha_g1=alpha_g1, beta_g1=beta_g1, delta_g1=delta_g1, beta_g2=beta_g2, gamma_g2=gamma_g2, delta_g2=delta_g2,
                      gamma_abc_g1=gamma_abc, a_query=a_q, b_g1_query=b1_q, b_g2_query=b2_q, h_query=h_q, l_query=l_q)


def proof_to_bytes_abi(ctx, curve, proof_limbs, inf) -> bytes:
    """the 3 affine Montgomery points zkp_groth16_prove returns (A | B | C) -> `Proof::serialize` bytes (cli/src/zkp_prove.rs:45-49)"""
    import numpy as np
    c = get_curve(curve)
    w = 2 * c.fq_limbs
    p = np.ascontiguousarray(proof_limbs, dtype=np.uint64).reshape(-1)
    return (ctx.compress_points(c, 1, p[:w].reshape(1, -1), [inf[0]]) + ctx.compress_points(c, 2, p[w:3 * w].reshape(1, -1), [inf[1]]) +
            ctx.compress_points(c, 1, p[3 * w:4 * w].reshape(1, -1), [inf[2]]))
