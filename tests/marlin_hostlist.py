"""TEST INFRASTRUCTURE (moved out of the product package in round 3): a second, host-list implementation of the Marlin prover
over the C ABI, kept only to cross-check the device-resident product prover (ckb_zkp_amd/marlin.py -> csrc/marlin.hip).

Host-side mirror of the reference's Marlin prover on the MI355X backend (BASELINE.json configs[3]).

Mirrors /root/reference/marlin/src/lib.rs:97-181 (`create_random_proof`), ahp/indexer.rs:70-117 (`AHP::index`),
ahp/prover.rs:86-427 (`prover_init`, `prover_{first,second,third}_round`) and pc/mod.rs:34-160 (`PC::commit`,
`open`, `batch_open`).  Every NTT (interpolate / fft / evaluate_over_domain), element-wise product, batch inversion,
polynomial evaluation, witness-polynomial division and every KZG10 MSM runs on the device through the C ABI; the
O(n) glue between them (vanishing-polynomial folds, sparse accumulation of t, re-indexing) is host code for now.

The Fiat–Shamir transcript (merlin + ChaCha20, fs_rng.rs) is not reproduced: challenges and all
prover randomness are explicit inputs, which is what makes round-by-round bit-exact parity with the oracle possible.
"""
from __future__ import annotations

import numpy as np

from ckb_zkp_amd import api, codec, kzg10
from ckb_zkp_amd.api import Context
from ckb_zkp_amd.marlin import (INDEX_LABELS, LABELS_1, LABELS_2, LABELS_3, ZK_BOUND, MarlinCS, _next_pow2,  # noqa: F401
                                    index_matrices, reindex_by_subdomain)
from ckb_zkp_amd.params import get_curve


def _trim(p):
    p = list(p)
    while p and p[-1] == 0:
        p.pop()
    return p


class Dev:
    """Device-backed field-vector operations on lists of canonical ints (upload -> kernel -> download)."""

    def __init__(self, ctx: Context, curve):
        self.ctx, self.c = ctx, get_curve(curve)
        self.r = self.c.r

    def _m(self, xs):
        return codec.fr_to_mont(xs, self.c).reshape(-1, 4)

    def ntt(self, xs, size, op):
        a = self._m(list(xs) + [0] * (size - len(xs)))
        return codec.fr_from_mont(self.ctx.ntt(self.c, a, op), self.c)

    def fft(self, coeffs, size):
        return self.ntt(_trim(coeffs), size, api.NTT_FFT)

    def ifft(self, evals, size=None):
        """EvaluationsOnDomain::interpolate — evaluations are zero-padded to the domain size like ark-poly does."""
        return self.ntt(evals, size or _next_pow2(len(evals)), api.NTT_IFFT)

    def vec(self, op, a, b=None, k=None):
        n = len(a)
        da = self.ctx.to_device(self._m(a))
        db = self.ctx.to_device(self._m(b)) if b is not None else None
        try:
            self.ctx.fr_vec_op(self.c, op, da, db, da, n, None if k is None else codec.fr_to_mont([k], self.c)[0])
            out = np.zeros((n, 4), dtype=np.uint64)
            self.ctx.d2h(out, da)
        finally:
            self.ctx.dev_free(da)
            if db:
                self.ctx.dev_free(db)
        return codec.fr_from_mont(out, self.c)

    def batch_inverse(self, xs):
        d = self.ctx.to_device(self._m(xs))
        try:
            self.ctx.fr_batch_inverse(self.c, d, len(xs))
            out = np.zeros((len(xs), 4), dtype=np.uint64)
            self.ctx.d2h(out, d)
        finally:
            self.ctx.dev_free(d)
        return codec.fr_from_mont(out, self.c)

    def evaluate(self, coeffs, z):
        if not coeffs:
            return 0
        d = self.ctx.to_device(self._m(coeffs))
        try:
            v = self.ctx.poly_evaluate(self.c, d, len(coeffs), codec.fr_to_mont([z], self.c)[0])
        finally:
            self.ctx.dev_free(d)
        return codec.fr_from_mont(v.reshape(1, 4), self.c)[0]

    def pmul(self, a, b):
        """DensePolynomial * DensePolynomial via NTT on next_pow2(len_a + len_b - 1)."""
        a, b = _trim(a), _trim(b)
        if not a or not b:
            return []
        size = _next_pow2(len(a) + len(b) - 1)
        return _trim(self.ifft(self.vec(api.VEC_MUL, self.fft(a, size), self.fft(b, size))))


# ---- O(n) host glue
def _padd(a, b, r, kb=1):
    n = max(len(a), len(b))
    return [((a[i] if i < len(a) else 0) + kb * (b[i] if i < len(b) else 0)) % r for i in range(n)]


def divide_by_vanishing(p, n, r):
    p = list(p)
    if len(p) <= n:
        return [], _trim(p)
    q = p[n:]
    for i in range(len(q) - 1, -1, -1):
        if i + n < len(q):
            q[i] = (q[i] + q[i + n]) % r
    return _trim(q), _trim([(p[i] + (q[i] if i < len(q) else 0)) % r for i in range(n)])


def mul_by_vanishing(p, n, r):
    out = [0] * (len(p) + n)
    for i, c in enumerate(p):
        out[i + n] = (out[i + n] + c) % r
        out[i] = (out[i] - c) % r
    return out


def _domain(c, n):
    size = _next_pow2(n)
    lg = size.bit_length() - 1
    w = pow(pow(c.fr_generator, (c.r - 1) >> c.two_adicity, c.r), 1 << (c.two_adicity - lg), c.r)
    els, p = [], 1
    for _ in range(size):
        els.append(p)
        p = p * w % c.r
    return size, els


def index(ctx: Context, curve, circuit):
    """AHP::index (indexer.rs:70-117) + compose_matrix_polynomials (arithmetic.rs:98-172)."""
    c = get_curve(curve)
    r = c.r
    dev = Dev(ctx, c)
    cs, mats = index_matrices(c, circuit)
    a, b, cc = mats
    nnz = max(sum(map(len, m)) for m in mats)
    nvars = cs.num_inputs + cs.num_aux
    xs, _ = _domain(c, cs.num_inputs)
    hs, h_el = _domain(c, nvars)
    ks, _ = _domain(c, nnz)
    bs = _next_pow2(3 * ks - 3)
    diag_inv = dev.batch_inverse([hs * pow(e, -1, r) % r for e in h_el])

    def compose(m):
        row, col, val = [], [], []
        for i, rw in enumerate(m):
            for v, j in rw:
                jj = reindex_by_subdomain(hs, xs, j)
                row.append(h_el[jj])
                col.append(h_el[i])
                val.append(v * diag_inv[jj] % r)
        pad = ks - len(row)
        row += [h_el[0]] * pad
        col += [h_el[0]] * pad
        val += [0] * pad
        rc = dev.vec(api.VEC_MUL, row, col)
        polys = {k: dev.ifft(v) for k, v in (("row", row), ("col", col), ("val", val), ("row_col", rc))}
        return dict(polys=polys, on_k=dict(row=row, col=col, val=val), on_b={k: dev.fft(p, bs) for k, p in polys.items()})

    return dict(curve=c, num_inputs=cs.num_inputs, num_constraints=cs.num_constraints(), num_variables=nvars,
                num_non_zeros=nnz, a=a, b=b, c=cc, xs=xs, hs=hs, ks=ks, bs=bs, h_el=h_el,
                star=dict(a=compose(a), b=compose(b), c=compose(cc)),
                max_degree=max(3 * hs + 2 * ZK_BOUND - 1, 3 * ks - 3))


def create_proof(ctx: Context, idx, ck: kzg10.CommitterKey, circuit, rnd, ch):
    """create_random_proof with explicit randomness and challenges.
    rnd: w, z_a, z_b (1 coefficient each: the zk masks of prover.rs:190,196,200), mask (3|H| coefficients, :202-205),
         blind[label] / blind_shifted[label] (2 coefficients each: `Rand::rand(hiding_bound = 1)` of KZG10::commit);
    ch:  alpha, eta_a, eta_b, eta_c, beta, gamma (verifier messages) and xi (the opening challenge).
    Returns commitments, evaluations (query-set order) and the two opening proofs."""
    c = idx["curve"]
    r = c.r
    dev = Dev(ctx, c)
    xs, hs, ks, bs, h_el = idx["xs"], idx["hs"], idx["ks"], idx["bs"], idx["h_el"]
    # ---- prover_init
    cs = MarlinCS(c, assign=True)
    circuit.generate_constraints(cs)
    cs.make_matrices_square()
    x, w = cs.input_assignment, cs.aux_assignment
    z = x + w
    z_a_ev = [sum(cf * z[j] for cf, j in row) % r for row in idx["a"]]
    z_b_ev = [sum(cf * z[j] for cf, j in row) % r for row in idx["b"]]
    # ---- first round
    x_poly = dev.ifft(x + [0] * (xs - len(x)))
    x_on_h = dev.fft(x_poly, hs)
    ratio = hs // xs
    w_ext = w + [0] * (hs - xs - len(w))
    w_on_h = [0 if i % ratio == 0 else (w_ext[i - i // ratio - 1] - x_on_h[i]) % r for i in range(hs)]
    w_poly = _padd(dev.ifft(w_on_h), mul_by_vanishing(rnd["w"], hs, r), r)
    w_poly, rem = divide_by_vanishing(w_poly, xs, r)
    assert not rem
    z_a = _trim(_padd(dev.ifft(z_a_ev), mul_by_vanishing(rnd["z_a"], hs, r), r))
    z_b = _trim(_padd(dev.ifft(z_b_ev), mul_by_vanishing(rnd["z_b"], hs, r), r))
    mask = list(rnd["mask"])
    _, rem = divide_by_vanishing(mask, hs, r)
    mask[0] = (mask[0] - (rem[0] if rem else 0)) % r
    polys = {f"{m}_{k}": idx["star"][m]["polys"][k] for m in "abc" for k in ("row", "col", "val", "row_col")}
    polys.update(w=_trim(w_poly), z_a=z_a, z_b=z_b, mask=_trim(mask))
    # ---- second round
    alpha, ea, eb, ec, beta = ch["alpha"], ch["eta_a"], ch["eta_b"], ch["eta_c"], ch["beta"]
    zc = dev.pmul(z_a, z_b)
    m_poly = [(ec * zc[i] + ea * (z_a[i] if i < len(z_a) else 0) + eb * (z_b[i] if i < len(z_b) else 0)) % r
              for i in range(len(zc))]
    v_alpha = (pow(alpha, hs, r) - 1) % r
    r_alpha_on_h = dev.vec(api.VEC_SCALE, dev.batch_inverse([(alpha - u) % r for u in h_el]), k=v_alpha)
    r_alpha = dev.ifft(r_alpha_on_h)
    t_on_h = [0] * hs
    for mat, eta in ((idx["a"], ea), (idx["b"], eb), (idx["c"], ec)):
        for i, row in enumerate(mat):
            for cf, j in row:
                k = reindex_by_subdomain(hs, xs, j)
                t_on_h[k] = (t_on_h[k] + eta * cf % r * r_alpha_on_h[i]) % r
    t_poly = dev.ifft(t_on_h)
    z_poly = mul_by_vanishing(polys["w"], xs, r)
    for i, cf in enumerate(x_poly):
        z_poly[i] = (z_poly[i] + cf) % r
    size = _next_pow2(max(len(polys["mask"]), len(_trim(r_alpha)) + len(_trim(m_poly)), len(_trim(t_poly)) + len(_trim(z_poly))))
    re, me, te, ze = (dev.fft(p, size) for p in (r_alpha, m_poly, t_poly, z_poly))
    q1 = _padd(polys["mask"], dev.ifft(dev.vec(api.VEC_SUB, dev.vec(api.VEC_MUL, re, me), dev.vec(api.VEC_MUL, te, ze))), r)
    h1, xg1 = divide_by_vanishing(q1, hs, r)
    polys.update(t=_trim(t_poly), g_1=_trim(xg1[1:]), h_1=_trim(h1))
    # ---- third round
    va, vb = v_alpha, (pow(beta, hs, r) - 1) % r
    S = idx["star"]
    inv = {}
    for nm in "abc":
        bm = [(beta - rw) % r for rw in S[nm]["on_k"]["row"]]
        am = [(alpha - cl) % r for cl in S[nm]["on_k"]["col"]]
        inv[nm] = dev.batch_inverse(dev.vec(api.VEC_MUL, bm, am))
    acc = [0] * ks
    for nm, eta in (("a", ea), ("b", eb), ("c", ec)):
        acc = dev.vec(api.VEC_AXPY, acc, dev.vec(api.VEC_MUL, S[nm]["on_k"]["val"], inv[nm]), k=eta)
    t3 = dev.ifft(dev.vec(api.VEC_SCALE, acc, k=va * vb % r))
    den = {}
    ab = alpha * beta % r
    for nm in "abc":
        d = dev.vec(api.VEC_AXPY, S[nm]["on_b"]["row_col"], S[nm]["on_b"]["row"], k=(-alpha) % r)
        d = dev.vec(api.VEC_AXPY, d, S[nm]["on_b"]["col"], k=(-beta) % r)
        den[nm] = [(v + ab) % r for v in d]
    prod = lambda u, v: dev.vec(api.VEC_MUL, u, v)
    a_on_b = [0] * bs
    for nm, eta, o1_, o2_ in (("a", ea, "b", "c"), ("b", eb, "c", "a"), ("c", ec, "a", "b")):
        a_on_b = dev.vec(api.VEC_AXPY, a_on_b, prod(prod(S[nm]["on_b"]["val"], den[o1_]), den[o2_]), k=eta)
    a_poly = dev.ifft(dev.vec(api.VEC_SCALE, a_on_b, k=va * vb % r))
    b_poly = dev.ifft(prod(prod(den["a"], den["b"]), den["c"]))
    h2, _ = divide_by_vanishing(_padd(a_poly, dev.pmul(b_poly, t3), r, kb=-1), ks, r)
    polys.update(g_2=_trim(t3[1:]), h_2=_trim(h2))
    # ---- PC::commit (pc/mod.rs:34-71)
    D = idx["max_degree"]
    bounds = {"g_1": hs - 2, "g_2": ks - 2}
    hide = lambda l: l in ("w", "z_a", "z_b", "g_1")
    mont = lambda p: codec.fr_to_mont(p, c).reshape(-1, 4)
    blind = lambda l: mont(rnd["blind"][l]) if hide(l) else None
    blind_s = lambda l: mont(rnd["blind_shifted"][l]) if (hide(l) and l in bounds) else None
    comms = {}
    for l in LABELS_1 + LABELS_2 + LABELS_3:
        comm = kzg10.commit(ctx, ck, mont(polys[l]), blind(l))
        shifted = None
        if l in bounds:
            shifted = kzg10.commit(ctx, ck, mont(polys[l]), blind_s(l), power_offset=D - bounds[l])
        comms[l] = (comm, shifted)
    # ---- evaluations + batch_open (lib.rs:147-165, pc/mod.rs:73-160)
    query = sorted([(l, beta) for l in LABELS_1 + LABELS_2] + [(l, ch["gamma"]) for l in LABELS_3 + INDEX_LABELS])
    evals = [dev.evaluate(polys[l], pt) for l, pt in query]
    xi = ch["xi"]
    proofs = []
    for pt in sorted({pt for _, pt in query}):
        p, rb, chal = [], [], 1
        for l in sorted(l for l, q in query if q == pt):
            p = _padd(p, polys[l], r, kb=chal)
            if hide(l):
                rb = _padd(rb, rnd["blind"][l], r, kb=chal)
            if l in bounds:
                sc = chal * xi % r
                p = _padd(p, [0] * (D - bounds[l]) + polys[l], r, kb=sc)
                if hide(l):
                    rb = _padd(rb, rnd["blind_shifted"][l], r, kb=sc)
            chal = chal * xi % r * xi % r
        proofs.append(kzg10.open(ctx, ck, mont(p), pt, mont(rb) if any(rb) else None))
    return dict(commitments=comms, evaluations=evals, opening_proofs=proofs, polys=polys, query=query)
