"""ORACLE (test infrastructure only) — Marlin's Fiat–Shamir RNG in plain Python.

Follows /root/reference/marlin/src/fs_rng.rs:11-70 (`FiatShamirRng`: merlin transcript -> 32-byte seed -> ChaChaRng) and
the sampling calls of /root/reference/marlin/src/lib.rs:105-158, ahp/verifier.rs:41-87,118-127.

Third-party algorithms the reference pulls from crates that are NOT vendored (Cargo "2.0" / "0.2" / "0.7", no lock file):
restated here from their published specifications and pinned as far as this container allows
  * Keccak-f[1600]           FIPS 202 — pinned: SHA3-256 built on this permutation == hashlib.sha3_256 (self_check);
  * STROBE-128 / merlin 2.0  "Merlin v1.0" transcripts over STROBEv1.0.2 — pinned against two published vectors
                             (merlin's strobe conformance test and the "test protocol" transcript vector, both RECALLED,
                             not read from a file: a wrong recollection cannot match a from-scratch implementation by
                             accident, so agreement pins both);
  * ChaCha20 (rand_chacha 0.2 `ChaChaRng` = 20 rounds, 64-bit block counter, zero stream id; rand_core 0.5 `BlockRng`
    buffering of 4 blocks)   pinned: all-zero key block 0 == the RFC 7539 / rand_chacha `test_chacha_true_values_a` words;
  * ark-ff 0.2 `UniformRand for Fp256` (4 x next_u64 little-endian limbs, top limb masked by REPR_SHAVE_BITS, rejection
    above the modulus, the accepted integer IS the Montgomery representation), rand 0.7 `u128` (low half first),
    ark `ToBytes` layouts (Fp: canonical integer little-endian; GroupAffine: x, y, infinity byte).  RECALLED — PARITY
    UNPINNED: the reference holds no transcript vector and cannot be built here.
"""
from __future__ import annotations

import hashlib
import struct

MASK64 = (1 << 64) - 1

# ------------------------------------------------------------------ Keccak-f[1600]
_RC = []
_r = 1
for _ in range(24):
    rc = 0
    for j in range(7):
        _r = ((_r << 1) ^ ((_r >> 7) * 0x71)) & 0xFF
        if _r & 2:
            rc ^= 1 << ((1 << j) - 1)
    _RC.append(rc)
# the LFSR above is the FIPS 202 rc(t) generator shifted by one step; assert the first / last constants
assert _RC[0] == 0x0000000000000001 and _RC[1] == 0x0000000000008082 and _RC[23] == 0x8000000080008008, [hex(x) for x in _RC[:2]]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & MASK64 if n else x


def keccak_f1600(state: bytearray) -> None:
    """in place on 200 bytes (lane (x, y) at byte offset 8 * (x + 5 y), little-endian)"""
    a = [[0] * 5 for _ in range(5)]
    for x in range(5):
        for y in range(5):
            a[x][y] = int.from_bytes(state[8 * (x + 5 * y):8 * (x + 5 * y) + 8], "little")
    for rnd in range(24):
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        for x in range(5):
            for y in range(5):
                a[x][y] ^= d[x]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        for x in range(5):
            for y in range(5):
                a[x][y] = b[x][y] ^ ((~b[(x + 1) % 5][y]) & MASK64 & b[(x + 2) % 5][y])
        a[0][0] ^= _RC[rnd]
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y):8 * (x + 5 * y) + 8] = a[x][y].to_bytes(8, "little")


def _sha3_256(msg: bytes) -> bytes:
    st, rate = bytearray(200), 136
    m = bytearray(msg) + b"\x06"
    m += b"\x00" * (-len(m) % rate)
    m[-1] |= 0x80
    for o in range(0, len(m), rate):
        for i in range(rate):
            st[i] ^= m[o + i]
        keccak_f1600(st)
    return bytes(st[:32])


# ------------------------------------------------------------------ STROBE-128 (the subset merlin uses)
FLAG_I, FLAG_A, FLAG_C, FLAG_T, FLAG_M, FLAG_K = 1, 2, 4, 8, 16, 32
STROBE_R = 166


class Strobe128:
    def __init__(self, protocol_label: bytes):
        st = bytearray(200)
        st[0:6] = bytes([1, STROBE_R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        keccak_f1600(st)
        self.state, self.pos, self.pos_begin, self.cur_flags = st, 0, 0, 0
        self.meta_ad(protocol_label, False)

    def _run_f(self):
        self.state[self.pos] ^= self.pos_begin
        self.state[self.pos + 1] ^= 0x04
        self.state[STROBE_R + 1] ^= 0x80
        keccak_f1600(self.state)
        self.pos = self.pos_begin = 0

    def _absorb(self, data: bytes):
        for b in data:
            self.state[self.pos] ^= b
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()

    def _overwrite(self, data: bytes):
        for b in data:
            self.state[self.pos] = b
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()

    def _squeeze(self, n: int) -> bytes:
        out = bytearray()
        for _ in range(n):
            out.append(self.state[self.pos])
            self.state[self.pos] = 0
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags: int, more: bool):
        if more:
            assert self.cur_flags == flags
            return
        assert not flags & FLAG_T
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        if flags & (FLAG_C | FLAG_K) and self.pos != 0:
            self._run_f()

    def meta_ad(self, data: bytes, more: bool):
        self._begin_op(FLAG_M | FLAG_A, more)
        self._absorb(data)

    def ad(self, data: bytes, more: bool):
        self._begin_op(FLAG_A, more)
        self._absorb(data)

    def prf(self, n: int, more: bool = False) -> bytes:
        self._begin_op(FLAG_I | FLAG_A | FLAG_C, more)
        return self._squeeze(n)

    def key(self, data: bytes, more: bool = False):
        self._begin_op(FLAG_A | FLAG_C, more)
        self._overwrite(data)


class MerlinTranscript:
    """merlin 2.0 `Transcript`"""

    def __init__(self, label: bytes):
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label: bytes, message: bytes):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(struct.pack("<I", len(message)), True)
        self.strobe.ad(message, False)

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(struct.pack("<I", n), True)
        return self.strobe.prf(n, False)


# ------------------------------------------------------------------ ChaCha20 RNG (rand_chacha 0.2 / rand_core 0.5)
def _chacha_block(key_words, counter: int, stream: int = 0):
    def rotl(v, n):
        return ((v << n) | (v >> (32 - n))) & 0xFFFFFFFF

    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + \
        [counter & 0xFFFFFFFF, (counter >> 32) & 0xFFFFFFFF, stream & 0xFFFFFFFF, (stream >> 32) & 0xFFFFFFFF]
    x = list(init)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 7)

    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & 0xFFFFFFFF for a, b in zip(x, init)]


class ChaChaRng:
    """`ChaChaRng::from_seed(seed)`: 4-block (64-word) buffer, word-granular consumption as rand_core's BlockRng."""
    BUF = 64

    def __init__(self, seed: bytes):
        assert len(seed) == 32
        self.key = struct.unpack("<8I", seed)
        self.counter = 0
        self.results = [0] * self.BUF
        self.index = self.BUF

    def _generate_and_set(self, index: int):
        out = []
        for k in range(4):
            out += _chacha_block(self.key, self.counter + k)
        self.counter += 4
        self.results, self.index = out, index

    def next_u32(self) -> int:
        if self.index >= self.BUF:
            self._generate_and_set(0)
        v = self.results[self.index]
        self.index += 1
        return v

    def next_u64(self) -> int:
        n, i = self.BUF, self.index
        if i < n - 1:
            self.index += 2
            return (self.results[i + 1] << 32) | self.results[i]
        if i >= n:
            self._generate_and_set(2)
            return (self.results[1] << 32) | self.results[0]
        x = self.results[n - 1]
        self._generate_and_set(1)
        return (self.results[0] << 32) | x

    def fill_bytes(self, n: int) -> bytes:
        out = bytearray()
        while len(out) < n:
            if self.index >= self.BUF:
                self._generate_and_set(0)
            avail = self.results[self.index:]
            take = min(4 * len(avail), n - len(out))
            words = (take + 3) // 4
            out += struct.pack(f"<{words}I", *avail[:words])[:take]
            self.index += words
        return bytes(out)


# ------------------------------------------------------------------ FiatShamirRng (marlin/src/fs_rng.rs)
class FiatShamirRng:
    def __init__(self, seed_material: bytes):
        """from_seed(&bytes) — fs_rng.rs:41-53"""
        t = MerlinTranscript(b"MARLINSEED")
        t.append_message(b"Seed", bytes(seed_material))
        self.seed = t.challenge_bytes(b"x", 32)
        self.r = ChaChaRng(self.seed)

    def absorb(self, material: bytes):
        """fs_rng.rs:57-69: seed = H(material || seed)"""
        t = MerlinTranscript(b"MARLINSEED")
        t.append_message(b"Seed", bytes(material) + self.seed)
        self.seed = t.challenge_bytes(b"x", 32)
        self.r = ChaChaRng(self.seed)

    # ---- sampling (what lib.rs / ahp/verifier.rs draw from the rng)
    def next_u64(self) -> int:
        return self.r.next_u64()

    def rand_fr(self, curve) -> int:
        """ark-ff 0.2 `Fp256::rand`: canonical value of the element whose MONTGOMERY limbs were sampled"""
        shave = 256 - curve.r.bit_length()
        while True:
            limbs = [self.r.next_u64() for _ in range(4)]
            limbs[3] &= MASK64 >> shave
            raw = sum(l << (64 * i) for i, l in enumerate(limbs))
            if raw < curve.r:
                return raw * pow(1 << 256, -1, curve.r) % curve.r

    def rand_u128(self) -> int:
        x = self.r.next_u64()
        y = self.r.next_u64()
        return (y << 64) | x

    def sample_outside_domain(self, curve, domain_size: int) -> int:
        """ahp/verifier.rs:118-127"""
        while True:
            t = self.rand_fr(curve)
            if pow(t, domain_size, curve.r) != 1:
                return t


# ------------------------------------------------------------------ ark `ToBytes` layouts used by the transcript
def fr_bytes(x: int, curve) -> bytes:
    return (x % curve.r).to_bytes(32, "little")


def fq_bytes(x: int, curve) -> bytes:
    return (x % curve.q).to_bytes(8 * ((curve.q.bit_length() + 63) // 64), "little")


def g1_bytes(p, curve) -> bytes:
    """GroupAffine::write: x, y, infinity (ark's zero() is (0, 1, true))"""
    if p is None:
        return fq_bytes(0, curve) + fq_bytes(1, curve) + b"\x01"
    return fq_bytes(p[0], curve) + fq_bytes(p[1], curve) + b"\x00"


def g2_bytes(p, curve) -> bytes:
    if p is None:
        return fq_bytes(0, curve) * 2 + fq_bytes(1, curve) + fq_bytes(0, curve) + b"\x01"
    (x0, x1), (y0, y1) = p
    return fq_bytes(x0, curve) + fq_bytes(x1, curve) + fq_bytes(y0, curve) + fq_bytes(y1, curve) + b"\x00"


def commitment_bytes(comm, curve) -> bytes:
    """pc/data_structures.rs:143-154: comm, shifted_exists (1 byte), shifted or the empty commitment"""
    c, s = comm
    return g1_bytes(c, curve) + (b"\x01" if s is not None else b"\x00") + g1_bytes(s, curve)


def index_verifier_key_bytes(ivk: dict, curve) -> bytes:
    """data_structures.rs:24-33 + indexer.rs:19-26 + pc/data_structures.rs:111-119.
    ivk: num_variables, num_constraints, num_non_zeros, index_comms [(comm, shifted)], g, gamma_g (G1), h, beta_h (G2),
    supported_degree"""
    out = struct.pack("<QQQ", ivk["num_variables"], ivk["num_constraints"], ivk["num_non_zeros"])
    out += struct.pack("<I", len(ivk["index_comms"]))
    for cm in ivk["index_comms"]:
        out += commitment_bytes(cm, curve)
    out += g1_bytes(ivk["g"], curve) + g1_bytes(ivk["gamma_g"], curve) + g2_bytes(ivk["h"], curve) + g2_bytes(ivk["beta_h"], curve)
    return out + struct.pack("<Q", ivk["supported_degree"])


def self_check():
    assert _sha3_256(b"") == hashlib.sha3_256(b"").digest()
    m = bytes(range(256)) * 3
    assert _sha3_256(m) == hashlib.sha3_256(m).digest()
    # merlin strobe.rs `test_conformance`
    s = Strobe128(b"Conformance Test Protocol")
    s.meta_ad(b"ms", False)
    s.meta_ad(b"g", True)
    s.ad(bytes([99]) * 1024, False)
    s.meta_ad(b"prf", False)
    p1 = s.prf(32)
    assert p1.hex() == "b48e645ca17c667fd5206ba57a6a228d72d8e1903814d3f17f622996d7cfefb0", p1.hex()
    s.meta_ad(b"key", False)
    s.key(p1)
    s.meta_ad(b"prf", False)
    p2 = s.prf(32)
    assert p2.hex() == "07e45cce8078cee259e3e375bb85d75610e2d1e1201c5f645045a194edd49ff8", p2.hex()
    # merlin transcript vector ("test protocol" / "some label" / "some data" / "challenge")
    t = MerlinTranscript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    # ChaCha20, all-zero key, block 0
    r = ChaChaRng(bytes(32))
    first = [r.next_u32() for _ in range(16)]
    assert first == [0xade0b876, 0x903df1a0, 0xe56a5d40, 0x28bd8653, 0xb819d2bd, 0x1aed8da0, 0xccef36a8, 0xc70d778b,
                     0x7c5941da, 0x8d485751, 0x3fe02477, 0x374ad8b8, 0xf4b8436a, 0x1ca11815, 0x69b687c3, 0x8665eeb2]
    return True
