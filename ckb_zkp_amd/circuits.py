"""Circuits the reference's own tests / examples use on this path.

* `Mini`      — /root/reference/groth16/tests/mini.rs:12-44  (x * (y + 2) = z, `num` copies, z public)
* `MimcChain` — /root/reference/marlin/examples/mimc.rs:15-119 (n independent MiMC-5 permutations; 12 aux and
                10 constraints per sample; the image is `alloc`ed, so the only input is the constant one).
* `mimc_chain_instance` — the same system written directly as arrays (no per-constraint closures) for the
  2^20 .. 2^24 benchmark instances of BASELINE.json (S = floor((2^k - 1)/10) samples -> domain 2^k).
"""
from __future__ import annotations

import numpy as np

from .codec import fr_to_mont
from .params import get_curve
from .r1cs import R1csInstance

MIMC_ROUNDS = 5


class Mini:
    def __init__(self, x=None, y=None, z=None, num=10):
        self.x, self.y, self.z, self.num = x, y, z, num

    def generate_constraints(self, cs):
        var_x = cs.alloc(lambda: self.x)
        var_y = cs.alloc(lambda: self.y)
        var_z = cs.alloc_input(lambda: self.z)
        for _ in range(self.num):
            cs.enforce(lambda lc: lc + var_x, lambda lc: lc + var_y + (2, cs.one()), lambda lc: lc + var_z)


class MimcChain:
    def __init__(self, curve, constants, preimages):
        self.r = get_curve(curve).r
        self.constants, self.preimages = list(constants), list(preimages)
        assert len(self.constants) == MIMC_ROUNDS

    def generate_constraints(self, cs):
        r = self.r
        for xl0, xr0 in self.preimages:
            xl_v = None if xl0 is None else xl0 % r
            xr_v = None if xr0 is None else xr0 % r
            xl = cs.alloc(lambda: xl_v)
            xr = cs.alloc(lambda: xr_v)
            for i in range(MIMC_ROUNDS):
                ci = self.constants[i]
                tmp_v = None if xl_v is None else (xl_v + ci) ** 2 % r
                tmp = cs.alloc(lambda: tmp_v)
                cs.enforce(lambda lc: lc + xl + (ci, cs.one()), lambda lc: lc + xl + (ci, cs.one()),
                           lambda lc: lc + tmp)
                new_v = None if xl_v is None else ((xl_v + ci) * tmp_v + xr_v) % r
                new_xl = cs.alloc(lambda: new_v)
                cs.enforce(lambda lc: lc + tmp, lambda lc: lc + xl + (ci, cs.one()), lambda lc: lc + new_xl - xr)
                xr, xr_v = xl, xl_v
                xl, xl_v = new_xl, new_v


def splitmix64(state: int):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def prf_field_elements(seed: int, count: int, modulus: int) -> list:
    """Counter-based PRF (SplitMix64 -> 256 bits -> rejection sample < modulus)."""
    out, st = [], seed & 0xFFFFFFFFFFFFFFFF
    bits = modulus.bit_length()
    mask = (1 << bits) - 1
    while len(out) < count:
        v = 0
        for k in range(4):
            st, w = splitmix64(st)
            v |= w << (64 * k)
        v &= mask
        if v < modulus:
            out.append(v)
    return out


def samples_for_domain(k: int) -> int:
    """S = floor((2^k - 1)/10): 10*S constraints + 1 input <= 2^k."""
    return ((1 << k) - 1) // 10


def mimc_chain_instance(curve, n_samples: int, seed: int = 0xC0FFEE, with_witness: bool = True) -> R1csInstance:
    """Array-form MiMC chain, identical (tests/test_host_r1cs.py) to synthesising `MimcChain` through
    `ConstraintSystem`.  Variable order per sample: xl, xr, then per round (tmp, new_xl)."""
    c = get_curve(curve)
    r = c.r
    vals = prf_field_elements(seed, MIMC_ROUNDS + 2 * n_samples, r)
    consts, pre = vals[:MIMC_ROUNDS], vals[MIMC_ROUNDS:]
    S = n_samples
    num_aux, nc = 12 * S, 10 * S
    # aux indices (0-based within aux) for sample s: base = 12 s; xl0 = base, xr0 = base+1, tmp_i = base+2+2i,
    # new_i = base+3+2i.  Column index in z = 1 + aux index (one input).
    base = 12 * np.arange(S, dtype=np.int64)
    a_cols, a_ptr, b_cols, c_cols = [], [], [], []
    # per round index arrays
    xl_idx = [base] + [base + 3 + 2 * i for i in range(MIMC_ROUNDS - 1)]             # xl at round i
    xr_idx = [base + 1, base] + [base + 3 + 2 * i for i in range(MIMC_ROUNDS - 2)]   # xr at round i
    tmp_idx = [base + 2 + 2 * i for i in range(MIMC_ROUNDS)]
    new_idx = [base + 3 + 2 * i for i in range(MIMC_ROUNDS)]
    one_m = fr_to_mont([1], c)[0]
    neg_one_m = fr_to_mont([r - 1], c)[0]
    const_m = fr_to_mont(consts, c)
    # Row layout: constraint (s, i, 0): A = xl + C_i*one, B = same, C = tmp ; (s, i, 1): A = tmp, B = xl + C_i*one,
    # C = new_xl - xr.  nnz per row: (2,2,1) then (1,2,2).
    rows = nc
    a_nnz = np.tile(np.array([2, 1], dtype=np.uint32), rows // 2)
    b_nnz = np.full(rows, 2, dtype=np.uint32)
    c_nnz = np.tile(np.array([1, 2], dtype=np.uint32), rows // 2)

    def ptr(nnz):
        p = np.zeros(rows + 1, dtype=np.uint32)
        np.cumsum(nnz, out=p[1:])
        return p

    a_col = np.zeros((S, MIMC_ROUNDS, 3), dtype=np.uint32)
    a_cf = np.zeros((S, MIMC_ROUNDS, 3, 4), dtype=np.uint64)
    b_col = np.zeros((S, MIMC_ROUNDS, 4), dtype=np.uint32)
    b_cf = np.zeros((S, MIMC_ROUNDS, 4, 4), dtype=np.uint64)
    c_col = np.zeros((S, MIMC_ROUNDS, 3), dtype=np.uint32)
    c_cf = np.zeros((S, MIMC_ROUNDS, 3, 4), dtype=np.uint64)
    for i in range(MIMC_ROUNDS):
        xl = (1 + xl_idx[i]).astype(np.uint32)
        xr = (1 + xr_idx[i]).astype(np.uint32)
        tmp = (1 + tmp_idx[i]).astype(np.uint32)
        new = (1 + new_idx[i]).astype(np.uint32)
        # A: [xl, one*C] [tmp]
        a_col[:, i, 0], a_col[:, i, 1], a_col[:, i, 2] = xl, 0, tmp
        a_cf[:, i, 0], a_cf[:, i, 1], a_cf[:, i, 2] = one_m, const_m[i], one_m
        # B: [xl, one*C] [xl, one*C]
        b_col[:, i, 0], b_col[:, i, 1], b_col[:, i, 2], b_col[:, i, 3] = xl, 0, xl, 0
        b_cf[:, i, 0], b_cf[:, i, 1], b_cf[:, i, 2], b_cf[:, i, 3] = one_m, const_m[i], one_m, const_m[i]
        # C: [tmp] [new, -xr]
        c_col[:, i, 0], c_col[:, i, 1], c_col[:, i, 2] = tmp, new, xr
        c_cf[:, i, 0], c_cf[:, i, 1], c_cf[:, i, 2] = one_m, one_m, neg_one_m
    csr_a = (ptr(a_nnz), a_col.reshape(-1), a_cf.reshape(-1, 4))
    csr_b = (ptr(b_nnz), b_col.reshape(-1), b_cf.reshape(-1, 4))
    csr_c = (ptr(c_nnz), c_col.reshape(-1), c_cf.reshape(-1, 4))
    z = None
    if with_witness:
        z = [1]
        for s in range(S):
            xl_v, xr_v = pre[2 * s], pre[2 * s + 1]
            z.append(xl_v)
            z.append(xr_v)
            for i in range(MIMC_ROUNDS):
                t = (xl_v + consts[i]) % r
                tmp_v = t * t % r
                new_v = (t * tmp_v + xr_v) % r
                z.append(tmp_v)
                z.append(new_v)
                xr_v, xl_v = xl_v, new_v
    inst = R1csInstance(c, 1, num_aux, nc, csr_a, csr_b, csr_c, z)
    inst.constants, inst.preimages = consts, [(pre[2 * s], pre[2 * s + 1]) for s in range(S)]
    return inst


def boolean_mimc_instance(curve, log_n: int, seed: int = 0xB001EA4, with_witness: bool = True) -> R1csInstance:
    """SURVEY.md §8(d) "skewed variant": a satisfiable system of domain 2^log_n in which HALF of the aux variables are
    booleans (the witness shape of SHA / range-check circuits): S' MiMC-5 samples (12 aux, 10 constraints each) followed by
    nb = 12 S' bit variables, each with the booleanity row of the reference's `AllocatedBit::alloc`
    (/root/reference/gadgets/src/algebra/boolean.rs:67-90): (1 - b) * b = 0, i.e. A = one - b, B = b, C = empty —
    22 S' constraints + 1 input <= 2^log_n.  The bits come from the same counter PRF.  Half of the scalars of the A / B / L
    MSMs are 0 or 1: half of them vanish and one bucket of the first window receives a quarter of all entries — the case the
    0/1 fast paths and the bucket balancing exist for."""
    c = get_curve(curve)
    S = ((1 << log_n) - 1) // 22
    base = mimc_chain_instance(c, S, seed=seed, with_witness=with_witness)
    nb = 12 * S
    one_m, neg_one_m = fr_to_mont([1], c)[0], fr_to_mont([c.r - 1], c)[0]
    var = (1 + 12 * S + np.arange(nb)).astype(np.uint32)

    def ext(csr, nnz_per_row, new_cols, new_cf):
        ptr, col, cf = csr
        p2 = np.concatenate([ptr, ptr[-1] + nnz_per_row * (1 + np.arange(nb, dtype=np.int64))]).astype(np.uint32)
        return p2, np.concatenate([col, new_cols.astype(np.uint32)]), np.concatenate([cf, new_cf.reshape(-1, 4)])

    a_cols = np.stack([np.zeros(nb, dtype=np.uint32), var], axis=1).reshape(-1)           # one - b
    a_cf = np.tile(np.stack([one_m, neg_one_m]), (nb, 1))
    csr_a = ext(base.csr("a"), 2, a_cols, a_cf)
    csr_b = ext(base.csr("b"), 1, var, np.tile(one_m, (nb, 1)))
    csr_c = ext(base.csr("c"), 0, np.zeros(0, dtype=np.uint32), np.zeros((0, 4), dtype=np.uint64))
    z = None
    if with_witness:
        st, bits = (seed ^ 0xB175) & 0xFFFFFFFFFFFFFFFF, []
        while len(bits) < nb:
            st, w = splitmix64(st)
            bits.extend((w >> k) & 1 for k in range(64))
        z = base.z + bits[:nb]
    inst = R1csInstance(c, 1, 12 * S + nb, 10 * S + nb, csr_a, csr_b, csr_c, z)
    inst.num_boolean = nb
    return inst
