"""Marlin's Fiat–Shamir RNG — host-side mirror of /root/reference/marlin/src/fs_rng.rs:11-70 over the library's C ABI
(`zkp_fs_rng_*`: merlin "MARLINSEED" transcript -> ChaCha20 stream, implemented in csrc/fs_rng.cpp), plus the arkworks
`ToBytes` layouts of what `create_random_proof` feeds it (marlin/src/lib.rs:105-158):

    FiatShamirRng.from_seed(to_bytes![index_verifier_key, public_input])      lib.rs:105-106
    absorb(to_bytes![first_comms]) ... absorb(&evaluations)                     lib.rs:112,120,127,157
    Fr::rand / sample_element_outside_domain / u128::rand                      ahp/verifier.rs:41-87,118-127; lib.rs:158

`ToBytes` (ark-ff / ark-ec 0.2, restated): Fp -> canonical integer, little-endian u64 limbs; GroupAffine -> x, y, one
infinity byte (identity = (0, 1, 1)); Fq2 -> c0, c1; u64 / u32 little-endian; bool one byte; Vec<T> -> elements back to
back, no length prefix.  The composite layouts are the reference's own impls (data_structures.rs:24-33,
pc/data_structures.rs:111-119,143-154, ahp/indexer.rs:19-26).  PARITY UNPINNED at the ark-primitive level (no Rust
toolchain, no transcript fixture in the reference); merlin / ChaCha20 themselves are pinned by published vectors
(tests/test_fs_rng.py).
"""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from . import _lib
from .codec import fr_from_mont
from .params import get_curve


class FiatShamirRng:
    def __init__(self, seed_material: bytes):
        self.lib = _lib.load()
        h = C.c_void_p()
        buf = (C.c_uint8 * max(len(seed_material), 1)).from_buffer_copy(seed_material or b"\x00")
        _lib.check(self.lib.zkp_fs_rng_new(C.cast(buf, C.c_void_p), len(seed_material), C.byref(h)), "zkp_fs_rng_new")
        self.h = h

    @classmethod
    def from_seed(cls, seed_material: bytes) -> "FiatShamirRng":
        return cls(seed_material)

    def absorb(self, material: bytes):
        buf = (C.c_uint8 * max(len(material), 1)).from_buffer_copy(material or b"\x00")
        _lib.check(self.lib.zkp_fs_rng_absorb(self.h, C.cast(buf, C.c_void_p), len(material)), "zkp_fs_rng_absorb")

    @property
    def seed(self) -> bytes:
        out = (C.c_uint8 * 32)()
        _lib.check(self.lib.zkp_fs_rng_seed(self.h, C.cast(out, C.c_void_p)), "zkp_fs_rng_seed")
        return bytes(out)

    def next_u64(self) -> int:
        v = C.c_uint64()
        _lib.check(self.lib.zkp_fs_rng_next_u64(self.h, C.byref(v)), "zkp_fs_rng_next_u64")
        return v.value

    def rand_fr(self, curve) -> int:
        """`Fr::rand(&mut fs_rng)` -> canonical integer"""
        c = get_curve(curve)
        out = np.zeros(4, dtype=np.uint64)
        _lib.check(self.lib.zkp_fs_rng_rand_fr(self.h, c.cid, C.c_void_p(out.ctypes.data)), "zkp_fs_rng_rand_fr")
        return fr_from_mont(out.reshape(1, 4), c)[0]

    def sample_outside_domain(self, curve, domain_size: int) -> int:
        c = get_curve(curve)
        assert domain_size & (domain_size - 1) == 0
        out = np.zeros(4, dtype=np.uint64)
        _lib.check(self.lib.zkp_fs_rng_sample_outside_domain(self.h, c.cid, domain_size.bit_length() - 1,
                                                             C.c_void_p(out.ctypes.data)), "zkp_fs_rng_sample_outside_domain")
        return fr_from_mont(out.reshape(1, 4), c)[0]

    def rand_u128(self) -> int:
        out = np.zeros(2, dtype=np.uint64)
        _lib.check(self.lib.zkp_fs_rng_rand_u128(self.h, C.c_void_p(out.ctypes.data)), "zkp_fs_rng_rand_u128")
        return int(out[0]) | (int(out[1]) << 64)

    def close(self):
        if getattr(self, "h", None):
            self.lib.zkp_fs_rng_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merlin_oneshot(label: bytes, msg_label: bytes, msg: bytes, chal_label: bytes, n: int) -> bytes:
    lib = _lib.load()
    out = (C.c_uint8 * n)()
    b = lambda x: C.cast((C.c_uint8 * max(len(x), 1)).from_buffer_copy(x or b"\x00"), C.c_void_p)
    _lib.check(lib.zkp_merlin_oneshot(b(label), len(label), b(msg_label), len(msg_label), b(msg), len(msg), b(chal_label),
                                      len(chal_label), C.cast(out, C.c_void_p), n), "zkp_merlin_oneshot")
    return bytes(out)


# ---------------------------------------------------------------- arkworks `ToBytes` layouts
def fr_bytes(x: int, curve) -> bytes:
    c = get_curve(curve)
    return (x % c.r).to_bytes(32, "little")


def _fq(x: int, c) -> bytes:
    return (x % c.q).to_bytes(8 * c.fq_limbs, "little")


def g1_bytes(p, curve) -> bytes:
    c = get_curve(curve)
    if p is None:
        return _fq(0, c) + _fq(1, c) + b"\x01"
    return _fq(p[0], c) + _fq(p[1], c) + b"\x00"


def g2_bytes(p, curve) -> bytes:
    c = get_curve(curve)
    if p is None:
        return _fq(0, c) + _fq(0, c) + _fq(1, c) + _fq(0, c) + b"\x01"
    (x0, x1), (y0, y1) = p
    return _fq(x0, c) + _fq(x1, c) + _fq(y0, c) + _fq(y1, c) + b"\x00"


def commitment_bytes(comm, curve) -> bytes:
    """pc::Commitment { comm, shifted_comm: Option } (pc/data_structures.rs:143-154)"""
    cm, shifted = comm
    return g1_bytes(cm, curve) + (b"\x01" if shifted is not None else b"\x00") + g1_bytes(shifted, curve)


def index_verifier_key_bytes(ivk: dict, curve) -> bytes:
    """IndexVerifierKey::write (data_structures.rs:24-33): index_info (3 x u64), u32 count, commitments, VerifierKey
    (g, gamma_g, h, beta_h, supported_degree as u64)"""
    out = struct.pack("<QQQ", ivk["num_variables"], ivk["num_constraints"], ivk["num_non_zeros"])
    out += struct.pack("<I", len(ivk["index_comms"]))
    for cm in ivk["index_comms"]:
        out += commitment_bytes(cm, curve)
    out += g1_bytes(ivk["g"], curve) + g1_bytes(ivk["gamma_g"], curve) + g2_bytes(ivk["h"], curve) + g2_bytes(ivk["beta_h"], curve)
    return out + struct.pack("<Q", ivk["supported_degree"])


# ---------------------------------------------------------------- verifier messages of create_random_proof
class FixedChallenger:
    """TEST HOOK ONLY: verifier messages supplied up front (round-by-round parity with the oracle).  A prover run this
    way is not sound — `FiatShamirChallenger` is what `create_random_proof` uses."""

    def __init__(self, ch: dict):
        self.ch = dict(ch)

    def first(self, comms):
        c = self.ch
        return c["alpha"], c["eta_a"], c["eta_b"], c["eta_c"]

    def second(self, comms):
        return self.ch["beta"]

    def third(self, comms):
        return self.ch["gamma"]

    def opening(self, evals):
        return self.ch["xi"]


class FiatShamirChallenger:
    """marlin/src/lib.rs:105-158: seed the rng with to_bytes![index_verifier_key, public_input]; before each verifier
    round absorb the round's commitments and draw that round's message (ahp/verifier.rs:41-87); absorb the evaluations and
    draw the 128-bit opening challenge (lib.rs:157-158)."""

    def __init__(self, curve, domain_h_size: int, ivk: dict, public_input):
        self.curve, self.hs = get_curve(curve), domain_h_size
        self.rng = FiatShamirRng(index_verifier_key_bytes(ivk, self.curve) +
                                 b"".join(fr_bytes(x, self.curve) for x in public_input))
        self.ch = {}

    def _absorb_comms(self, comms):
        self.rng.absorb(b"".join(commitment_bytes(cm, self.curve) for cm in comms))

    def first(self, comms):
        self._absorb_comms(comms)
        alpha = self.rng.sample_outside_domain(self.curve, self.hs)
        ea, eb, ec = (self.rng.rand_fr(self.curve) for _ in range(3))
        self.ch.update(alpha=alpha, eta_a=ea, eta_b=eb, eta_c=ec)
        return alpha, ea, eb, ec

    def second(self, comms):
        self._absorb_comms(comms)
        self.ch["beta"] = self.rng.sample_outside_domain(self.curve, self.hs)
        return self.ch["beta"]

    def third(self, comms):
        self._absorb_comms(comms)
        self.ch["gamma"] = self.rng.rand_fr(self.curve)
        return self.ch["gamma"]

    def opening(self, evals):
        self.rng.absorb(b"".join(fr_bytes(e, self.curve) for e in evals))
        self.ch["xi"] = self.rng.rand_u128()
        return self.ch["xi"]
