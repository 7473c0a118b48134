"""ORACLE (test infrastructure only) — Marlin prover (AHP + KZG10-based PC) in big-int Python.

Follows /root/reference/marlin/src:
  ahp/constraint_systems.rs:9-31,100-133   make_matrices_square, balance_matrices, reindex
  ahp/arithmetic.rs:18-46,98-172           BivariatePoly helpers, compose_matrix_polynomials
  ahp/indexer.rs:70-117                    AHP::index
  ahp/prover.rs:86-147,150-222,230-321,331-427   prover_init / first / second / third round
  ahp/verifier.rs:89-115,128-210           query set, verifier_equality_check
  pc/mod.rs:34-100,122-160,204-240         PC::commit / open / batch_open / accumulate_commitments_and_values
  lib.rs:97-181,184-250                    create_random_proof, verify_proof
What the reference draws from the zk RNG (masks, commitment blinders) is an explicit input.  The verifier messages
(alpha, eta_*, beta, gamma, opening challenge) come from a `challenger`: either FiatShamirChallenger — the reference's
transcript (fs_rng.rs; lib.rs:105-158: seed from index_verifier_key || public_input, absorb the round's commitments,
squeeze) via oracle/pyref/fs_rng.py — or FixedChallenger (a dict; per-round parity tests).
Polynomials are coefficient lists (low degree first) of canonical ints.
"""
from __future__ import annotations

from . import kzg10 as K
from .curves import Group
from .groth16 import AUX, INPUT, ConstraintSystem
from .ntt import Domain
from .pairing import Pairing

ZK_BOUND = 1


# ------------------------------------------------------------------ polynomial helpers
def trim(p):
    p = list(p)
    while p and p[-1] == 0:
        p.pop()
    return p


def padd(a, b, r, kb=1):
    n = max(len(a), len(b))
    return [((a[i] if i < len(a) else 0) + kb * (b[i] if i < len(b) else 0)) % r for i in range(n)]


def pscale(a, k, r):
    return [x * k % r for x in a]


def pmul(curve, a, b):
    """DensePolynomial * DensePolynomial: evaluate–multiply–interpolate on next_pow2(len_a + len_b - 1) (ark-poly)."""
    a, b = trim(a), trim(b)
    if not a or not b:
        return []
    d = Domain(curve, len(a) + len(b) - 1)
    ea, eb = d.fft(a), d.fft(b)
    return trim(d.ifft([x * y % curve.r for x, y in zip(ea, eb)]))


def divide_by_vanishing(p, n, r):
    """p = q (X^n - 1) + rem, deg rem < n."""
    p = list(p)
    if len(p) <= n:
        return [], trim(p)
    q = p[n:]
    for i in range(len(q) - 1, -1, -1):
        if i + n < len(q):
            q[i] = (q[i] + q[i + n]) % r
    rem = [(p[i] + (q[i] if i < len(q) else 0)) % r for i in range(n)]
    return trim(q), trim(rem)


def mul_by_vanishing(p, n, r):
    out = [0] * (len(p) + n)
    for i, c in enumerate(p):
        out[i + n] = (out[i + n] + c) % r
        out[i] = (out[i] - c) % r
    return out


def reindex_by_subdomain(h_size, x_size, j):
    period = h_size // x_size
    if j < x_size:
        return j * period
    i = j - x_size
    return i + i // (period - 1) + 1


# ------------------------------------------------------------------ indexer
class MarlinCS(ConstraintSystem):
    """IndexerConstraintSystem / ProverConstraintSystem (constraint_systems.rs)."""

    def make_matrices_square(self):
        nv, nc = self.num_inputs + self.num_aux, self.num_constraints()
        if nv < nc:
            for _ in range(nc - nv):
                self.alloc(lambda: 1)
        else:
            for _ in range(nv - nc):
                self.enforce(lambda lc: lc, lambda lc: lc, lambda lc: lc)


def _matrix(rows, num_inputs):
    return [[(coeff, j if kind == INPUT else num_inputs + j) for coeff, (kind, j) in row] for row in rows]


def _balance(a, b):
    da, db = sum(map(len, a)), sum(map(len, b))
    denser = da > db
    for i in range(len(a)):
        if denser:
            la, lb = len(a[i]), len(b[i])
            a[i], b[i] = b[i], a[i]
            da += lb - la
            db += la - lb
            denser = da > db


def index(curve, circuit):
    r = curve.r
    cs = MarlinCS(curve, want_values=False)
    circuit.generate_constraints(cs)
    cs.make_matrices_square()
    a, b, c = (_matrix(m, cs.num_inputs) for m in (cs.at, cs.bt, cs.ct))
    _balance(a, b)
    for m in (a, b, c):
        for row in m:
            row.sort(key=lambda t: t[1])
    nnz = max(sum(map(len, m)) for m in (a, b, c))
    nvars = cs.num_inputs + cs.num_aux
    dx, dh, dk = Domain(curve, cs.num_inputs), Domain(curve, nvars), Domain(curve, nnz)
    db = Domain(curve, 3 * dk.size - 3)
    h_el = dh.elements()
    n_h = dh.size % r
    diag = [n_h * pow(e, -1, r) % r for e in h_el]            # diagonal_evals: N * w^-i

    def compose(m):
        row, col, val = [], [], []
        for i, rw in enumerate(m):
            for v, j in rw:
                jj = reindex_by_subdomain(dh.size, dx.size, j)
                row.append(h_el[jj])
                col.append(h_el[i])
                val.append(v * pow(diag[jj], -1, r) % r)
        pad = dk.size - len(row)
        row += [h_el[0]] * pad
        col += [h_el[0]] * pad
        val += [0] * pad
        rc = [x * y % r for x, y in zip(row, col)]
        polys = {k: dk.ifft(v) for k, v in (("row", row), ("col", col), ("val", val), ("row_col", rc))}
        return dict(polys=polys, on_k=dict(row=row, col=col, val=val),
                    on_b={k: db.fft(p) for k, p in polys.items()})

    return dict(curve=curve, num_inputs=cs.num_inputs, num_constraints=cs.num_constraints(), num_variables=nvars,
                num_non_zeros=nnz, a=a, b=b, c=c, dx=dx, dh=dh, dk=dk, db=db,
                star=dict(a=compose(a), b=compose(b), c=compose(c)),
                max_degree=max(3 * dh.size + 2 * ZK_BOUND - 1, 3 * dk.size - 3))


# ------------------------------------------------------------------ prover rounds
def prover_init(idx, circuit):
    curve = idx["curve"]
    r = curve.r
    cs = MarlinCS(curve, want_values=True)
    circuit.generate_constraints(cs)
    cs.make_matrices_square()
    x, w = cs.input_assignment, cs.aux_assignment
    assert cs.num_constraints() == idx["num_constraints"] == len(x) + len(w)
    z = x + w
    ip = lambda row: sum(cf * z[j] for cf, j in row) % r
    return dict(x=x, w=w, z_a=[ip(rw) for rw in idx["a"]], z_b=[ip(rw) for rw in idx["b"]])


def first_round(idx, st, rnd):
    """rnd: dict(w=[1 coeff], z_a=[1], z_b=[1], mask=[3|H| coeffs])."""
    curve = idx["curve"]
    r = curve.r
    dx, dh = idx["dx"], idx["dh"]
    x_poly = dx.ifft(st["x"])
    x_on_h = dh.fft(x_poly)
    ratio = dh.size // dx.size
    w_ext = st["w"] + [0] * (dh.size - dx.size - len(st["w"]))
    w_on_h = [0 if i % ratio == 0 else (w_ext[i - i // ratio - 1] - x_on_h[i]) % r for i in range(dh.size)]
    vanish = lambda p: mul_by_vanishing(p, dh.size, r)
    w_poly = padd(dh.ifft(w_on_h), vanish(rnd["w"]), r)
    w_poly, rem = divide_by_vanishing(w_poly, dx.size, r)
    assert not rem
    z_a = padd(dh.ifft(st["z_a"]), vanish(rnd["z_a"]), r)
    z_b = padd(dh.ifft(st["z_b"]), vanish(rnd["z_b"]), r)
    mask = list(rnd["mask"])
    assert len(mask) == 3 * dh.size + 2 * ZK_BOUND - 2
    _, rem = divide_by_vanishing(mask, dh.size, r)
    mask[0] = (mask[0] - (rem[0] if rem else 0)) % r
    return dict(w=trim(w_poly), z_a=trim(z_a), z_b=trim(z_b), mask=trim(mask), x_poly=x_poly)


def second_round(idx, st, o1, alpha, eta_a, eta_b, eta_c):
    curve = idx["curve"]
    r = curve.r
    dx, dh = idx["dx"], idx["dh"]
    zc = pmul(curve, o1["z_a"], o1["z_b"])
    m = pscale(zc, eta_c, r)
    for i in range(len(m)):
        m[i] = (m[i] + eta_a * (o1["z_a"][i] if i < len(o1["z_a"]) else 0)
                + eta_b * (o1["z_b"][i] if i < len(o1["z_b"]) else 0)) % r
    v_alpha = dh.evaluate_vanishing_polynomial(alpha)
    r_alpha_on_h = [v_alpha * pow((alpha - u) % r, -1, r) % r for u in dh.elements()]    # batch_evals
    r_alpha = dh.ifft(r_alpha_on_h)
    t_on_h = [0] * dh.size
    for mat, eta in ((idx["a"], eta_a), (idx["b"], eta_b), (idx["c"], eta_c)):
        for i, row in enumerate(mat):
            for cf, j in row:
                k = reindex_by_subdomain(dh.size, dx.size, j)
                t_on_h[k] = (t_on_h[k] + eta * cf % r * r_alpha_on_h[i]) % r
    t = dh.ifft(t_on_h)
    z = mul_by_vanishing(o1["w"], dx.size, r)
    for i, c in enumerate(o1["x_poly"]):
        z[i] = (z[i] + c) % r
    size = max(len(o1["mask"]), len(trim(r_alpha)) + len(trim(m)), len(trim(t)) + len(trim(z)))
    d = Domain(curve, size)
    ev = lambda p: d.fft(trim(p))
    re, me, te, ze = ev(r_alpha), ev(m), ev(t), ev(z)
    q1 = padd(o1["mask"], d.ifft([(a * b - c * e) % r for a, b, c, e in zip(re, me, te, ze)]), r)
    h1, xg1 = divide_by_vanishing(q1, dh.size, r)
    assert not xg1 or xg1[0] == 0
    return dict(t=trim(t), g_1=trim(xg1[1:]), h_1=trim(h1))


def third_round(idx, alpha, eta_a, eta_b, eta_c, beta):
    curve = idx["curve"]
    r = curve.r
    dh, dk, db = idx["dh"], idx["dk"], idx["db"]
    va, vb = dh.evaluate_vanishing_polynomial(alpha), dh.evaluate_vanishing_polynomial(beta)
    S = idx["star"]
    inv = {}
    for nm in "abc":
        inv[nm] = [pow((beta - rw) * (alpha - cl) % r, -1, r) if (beta - rw) * (alpha - cl) % r else 0
                   for rw, cl in zip(S[nm]["on_k"]["row"], S[nm]["on_k"]["col"])]
    t_on_k = [(eta_a * S["a"]["on_k"]["val"][i] * inv["a"][i] + eta_b * S["b"]["on_k"]["val"][i] * inv["b"][i]
               + eta_c * S["c"]["on_k"]["val"][i] * inv["c"][i]) % r * va % r * vb % r for i in range(dk.size)]
    t_poly = dk.ifft(t_on_k)
    g_2 = trim(t_poly[1:])
    den = {nm: [(beta * alpha - alpha * rw - beta * cl + rc) % r for rw, cl, rc in
                zip(S[nm]["on_b"]["row"], S[nm]["on_b"]["col"], S[nm]["on_b"]["row_col"])] for nm in "abc"}
    a_on_b = [(eta_a * S["a"]["on_b"]["val"][i] * den["b"][i] * den["c"][i]
               + eta_b * S["b"]["on_b"]["val"][i] * den["c"][i] * den["a"][i]
               + eta_c * S["c"]["on_b"]["val"][i] * den["a"][i] * den["b"][i]) % r * va % r * vb % r
              for i in range(db.size)]
    b_on_b = [den["a"][i] * den["b"][i] % r * den["c"][i] % r for i in range(db.size)]
    a_poly, b_poly = db.ifft(a_on_b), db.ifft(b_on_b)
    h2, _ = divide_by_vanishing(padd(a_poly, pmul(curve, b_poly, t_poly), r, kb=-1), dk.size, r)
    return dict(g_2=g_2, h_2=trim(h2))


# ------------------------------------------------------------------ PC layer (pc/mod.rs)
def pc_commit(pp, supported_degree, poly, degree_bound=None, blind=None, blind_shifted=None):
    """-> (comm, shifted_comm or None)."""
    comm = K.commit(pp, poly, blind)
    shifted = None
    if degree_bound is not None:
        sp = dict(pp, powers_of_g=pp["powers_of_g"][supported_degree - degree_bound:])
        shifted = K.commit(sp, poly, blind_shifted)
    return comm, shifted


def pc_open(pp, supported_degree, items, point, xi):
    """items: [(poly, degree_bound, blind, blind_shifted)] in label order -> (w, rand_v)."""
    r = pp["curve"].r
    p, rb, ch = [], [], 1
    for poly, bound, blind, blind_s in items:
        p = padd(p, poly, r, kb=ch)
        if blind is not None:
            rb = padd(rb, blind, r, kb=ch)
        if bound is not None:
            sc = ch * xi % r
            p = padd(p, [0] * (supported_degree - bound) + list(poly), r, kb=sc)
            if blind_s is not None:
                rb = padd(rb, blind_s, r, kb=sc)
        ch = ch * xi % r * xi % r
    return K.open_(pp, p, point, rb if any(rb) else None)


def accumulate(pp, supported_degree, items, point, xi):
    """items: [(comm, shifted_comm, degree_bound, value)] -> (combined commitment, combined value)."""
    curve = pp["curve"]
    r = curve.r
    G1 = Group(curve, 1)
    cc, cv, ch = None, 0, 1
    for comm, shifted, bound, value in items:
        cc = G1.add(cc, G1.mul(comm, ch))
        cv = (cv + value * ch) % r
        if bound is not None:
            sc = ch * xi % r
            cc = G1.add(cc, G1.mul(shifted, sc))
            cv = (cv + pow(point, supported_degree - bound, r) * value % r * sc) % r
        ch = ch * xi % r * xi % r
    return cc, cv


# ------------------------------------------------------------------ create_proof / verify (lib.rs) with explicit challenges
LABELS_1, LABELS_2, LABELS_3 = ["w", "z_a", "z_b", "mask"], ["t", "g_1", "h_1"], ["g_2", "h_2"]
INDEX_LABELS = [f"{m}_{k}" for m in "abc" for k in ("row", "col", "val", "row_col")]


def degree_bounds(idx):
    return {"g_1": idx["dh"].size - 2, "g_2": idx["dk"].size - 2}


def hiding(label):
    return label in ("w", "z_a", "z_b", "g_1")


class FixedChallenger:
    """test hook: verifier messages supplied up front"""

    def __init__(self, ch):
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
    """lib.rs:105-158 (prover) == lib.rs:190-215 (verifier): FiatShamirRng::from_seed(to_bytes![ivk, public_input]);
    absorb(to_bytes![round commitments]) before each verifier round; absorb(&evaluations) before the opening challenge."""

    def __init__(self, idx, ivk: dict, public_input):
        from . import fs_rng as F
        self.F, self.curve, self.hs = F, idx["curve"], idx["dh"].size
        self.rng = F.FiatShamirRng(F.index_verifier_key_bytes(ivk, self.curve) +
                                   b"".join(F.fr_bytes(x, self.curve) for x in public_input))
        self.ch = {}

    def _absorb_comms(self, comms):
        self.rng.absorb(b"".join(self.F.commitment_bytes(cm, self.curve) for cm in comms))

    def first(self, comms):
        self._absorb_comms(comms)
        alpha = self.rng.sample_outside_domain(self.curve, self.hs)              # ahp/verifier.rs:53
        ea, eb, ec = (self.rng.rand_fr(self.curve) for _ in range(3))            # :54-56
        self.ch.update(alpha=alpha, eta_a=ea, eta_b=eb, eta_c=ec)
        return alpha, ea, eb, ec

    def second(self, comms):
        self._absorb_comms(comms)
        self.ch["beta"] = self.rng.sample_outside_domain(self.curve, self.hs)    # :76
        return self.ch["beta"]

    def third(self, comms):
        self._absorb_comms(comms)
        self.ch["gamma"] = self.rng.rand_fr(self.curve)                          # :86
        return self.ch["gamma"]

    def opening(self, evals):
        self.rng.absorb(b"".join(self.F.fr_bytes(e, self.curve) for e in evals))  # lib.rs:157
        self.ch["xi"] = self.rng.rand_u128()                                      # lib.rs:158
        return self.ch["xi"]


def index_verifier_key(idx, pp, index_comms) -> dict:
    """IndexVerifierKey (data_structures.rs:10-15) as the transcript sees it (lib.rs:79-84: comms in Index::iter order)"""
    return dict(num_variables=idx["num_variables"], num_constraints=idx["num_constraints"],
                num_non_zeros=idx["num_non_zeros"], index_comms=[index_comms[l] for l in INDEX_LABELS],
                g=pp["g"], gamma_g=pp["gamma_g"], h=pp["h"], beta_h=pp["beta_h"], supported_degree=idx["max_degree"])


def create_proof(idx, pp, circuit, rnd, ch):
    """lib.rs:97-181.  rnd: first-round masks + `blind[label]`, `blind_shifted[label]` (2 coeffs each, hiding bound 1);
    ch: a challenger (FiatShamirChallenger = the reference's create_random_proof) or a dict alpha, eta_a, eta_b, eta_c,
    beta, gamma, xi (FixedChallenger)."""
    chal = FixedChallenger(ch) if isinstance(ch, dict) else ch
    curve = idx["curve"]
    r = curve.r
    D = idx["max_degree"]
    bounds = degree_bounds(idx)
    blind = lambda l: rnd["blind"][l] if hiding(l) else None
    blind_s = lambda l: rnd["blind_shifted"][l] if (hiding(l) and l in bounds) else None
    polys = {f"{m}_{k}": idx["star"][m]["polys"][k] for m in "abc" for k in ("row", "col", "val", "row_col")}
    comms = {}

    def commit_round(labels):
        for l in labels:
            comms[l] = pc_commit(pp, D, polys[l], bounds.get(l), blind(l), blind_s(l))
        return [comms[l] for l in labels]

    st = prover_init(idx, circuit)
    o1 = first_round(idx, st, rnd)
    polys.update({k: o1[k] for k in LABELS_1})
    alpha, ea, eb, ec = chal.first(commit_round(LABELS_1))
    polys.update(second_round(idx, st, o1, alpha, ea, eb, ec))
    beta = chal.second(commit_round(LABELS_2))
    polys.update(third_round(idx, alpha, ea, eb, ec, beta))
    gamma = chal.third(commit_round(LABELS_3))
    query = sorted([(l, beta) for l in LABELS_1 + LABELS_2] + [(l, gamma) for l in LABELS_3 + INDEX_LABELS])
    evals = [K.evaluate(polys[l], pt, r) for l, pt in query]
    xi = chal.opening(evals)
    proofs = []
    for pt in sorted({pt for _, pt in query}):
        labels = sorted(l for l, p in query if p == pt)
        proofs.append(pc_open(pp, D, [(polys[l], bounds.get(l), blind(l), blind_s(l)) for l in labels], pt, xi))
    return dict(commitments=comms, evaluations=evals, opening_proofs=proofs, polys=polys, query=query,
                challenges=dict(alpha=alpha, eta_a=ea, eta_b=eb, eta_c=ec, beta=beta, gamma=gamma, xi=xi))


def create_random_proof(idx, pp, index_comms, circuit, rnd):
    """marlin::create_random_proof (lib.rs:97-181): verifier messages derived from the Fiat–Shamir transcript."""
    st = prover_init(idx, circuit)
    chal = FiatShamirChallenger(idx, index_verifier_key(idx, pp, index_comms), st["x"][1:])
    return create_proof(idx, pp, circuit, rnd, chal)


def verify_random_proof(idx, pp, index_comms, proof, public_input) -> bool:
    """marlin::verify_proof (lib.rs:184-250): replay the transcript on the proof's commitments / evaluations, then the
    algebraic and pairing checks.  proof needs only commitments, evaluations, opening_proofs (query is rebuilt)."""
    chal = FiatShamirChallenger(idx, index_verifier_key(idx, pp, index_comms), public_input)
    cm = proof["commitments"]
    chal.first([cm[l] for l in LABELS_1])
    beta = chal.second([cm[l] for l in LABELS_2])
    gamma = chal.third([cm[l] for l in LABELS_3])
    chal.opening(proof["evaluations"])
    query = sorted([(l, beta) for l in LABELS_1 + LABELS_2] + [(l, gamma) for l in LABELS_3 + INDEX_LABELS])
    return verify_proof(idx, pp, index_comms, dict(proof, query=query), public_input, chal.ch)


def index_commitments(idx, pp):
    return {f"{m}_{k}": (K.commit(pp, idx["star"][m]["polys"][k]), None) for m in "abc"
            for k in ("row", "col", "val", "row_col")}


def verify_proof(idx, pp, index_comms, proof, public_input, ch):
    """lib.rs:184-250 with the challenges supplied: verifier_equality_check + PC::batch_check (pairing)."""
    curve = idx["curve"]
    r = curve.r
    dh, dk = idx["dh"], idx["dk"]
    ev = {q: e for q, e in zip(proof["query"], proof["evaluations"])}
    alpha, beta, gamma = ch["alpha"], ch["beta"], ch["gamma"]
    ea, eb, ec = ch["eta_a"], ch["eta_b"], ch["eta_c"]
    va, vb = dh.evaluate_vanishing_polynomial(alpha), dh.evaluate_vanishing_polynomial(beta)
    r_ab = (va - vb) * pow((alpha - beta) % r, -1, r) % r if alpha != beta else dh.size * pow(alpha, dh.size - 1, r) % r
    fx = [1] + list(public_input)
    dx = Domain(curve, len(fx))
    x_at_beta = K.evaluate(dx.ifft(fx), beta, r)
    g = lambda l, p: ev[(l, p)]
    lhs = (g("mask", beta) + r_ab * (ea * g("z_a", beta) + eb * g("z_b", beta) + ec * g("z_a", beta) * g("z_b", beta))
           - g("t", beta) * (dx.evaluate_vanishing_polynomial(beta) * g("w", beta) + x_at_beta)) % r
    if lhs != (g("h_1", beta) * vb + beta * g("g_1", beta)) % r:
        return False
    den = {m: (alpha * beta - alpha * g(f"{m}_row", gamma) - beta * g(f"{m}_col", gamma) + g(f"{m}_row_col", gamma)) % r
           for m in "abc"}
    a_g = (ea * g("a_val", gamma) * den["b"] * den["c"] + eb * g("b_val", gamma) * den["c"] * den["a"]
           + ec * g("c_val", gamma) * den["a"] * den["b"]) % r * va % r * vb % r
    b_g = den["a"] * den["b"] * den["c"] % r
    rhs = (a_g - b_g * (gamma * g("g_2", gamma) + g("t", beta) * pow(dk.size, -1, r))) % r
    if g("h_2", gamma) * dk.evaluate_vanishing_polynomial(gamma) % r != rhs:
        return False
    comms = dict(index_comms)
    comms.update(proof["commitments"])
    bounds = degree_bounds(idx)
    D = idx["max_degree"]
    pts = sorted({pt for _, pt in proof["query"]})
    for pt, (w, rand_v) in zip(pts, proof["opening_proofs"]):
        labels = sorted(l for l, p in proof["query"] if p == pt)
        cc, cv = accumulate(pp, D, [(comms[l][0], comms[l][1], bounds.get(l), ev[(l, pt)]) for l in labels], pt, ch["xi"])
        if not K.check(pp, cc, pt, cv, w, rand_v):
            return False
    return True
