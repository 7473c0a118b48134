"""Host-side mirror of the reference's Marlin interface on the MI355X backend (/root/reference/marlin/src/lib.rs:69-181:
`index`, `create_random_proof`), over the C ABI: `NativeIndex` = zkp_marlin_index_upload (the arithmetization computed and kept
resident by the library), `prove_native` / `create_random_proof` = ONE zkp_marlin_prove call per proof (csrc/marlin.hip: AHP
rounds, PC::commit, Fiat-Shamir transcript, evaluations, batch_open — no Python between the rounds).

The host keeps what is closure-driven in the reference: synthesis and the index-manipulation half of AHP::index
(make_matrices_square, balance_matrices, per-row column sort: ahp/constraint_systems.rs:9-31,100-133) — `index_matrices` for
synthesizers, `prepare_matrices` for array-form instances.

(The Python-orchestrated device prover this module used to hold — every round a sequence of C-ABI vector calls — lives under
tests/marlin_pyorch.py since round 4: it is a cross-check of the library's prover, not a product path.)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib, api, codec, kzg10
from .api import Context
from .params import get_curve
from .r1cs import INPUT, ConstraintSystem

# ---- shared with the cross-check provers under tests/ (tests/marlin_hostlist.py, tests/marlin_pyorch.py)
ZK_BOUND = 1
LABELS_1, LABELS_2, LABELS_3 = ["w", "z_a", "z_b", "mask"], ["t", "g_1", "h_1"], ["g_2", "h_2"]
INDEX_LABELS = [f"{m}_{k}" for m in "abc" for k in ("row", "col", "val", "row_col")]


def _next_pow2(n):
    s = 1
    while s < n:
        s <<= 1
    return s


def reindex_by_subdomain(h_size, x_size, j):
    period = h_size // x_size
    if j < x_size:
        return j * period
    i = j - x_size
    return i + i // (period - 1) + 1


class MarlinCS(ConstraintSystem):
    """IndexerConstraintSystem / ProverConstraintSystem (ahp/constraint_systems.rs)."""

    def make_matrices_square(self):
        nv, nc = self.num_inputs + self.num_aux, self.num_constraints()
        if nv < nc:
            for _ in range(nc - nv):
                self.alloc(lambda: 1)
        else:
            for _ in range(nv - nc):
                self.enforce(lambda lc: lc, lambda lc: lc, lambda lc: lc)


def index_matrices(curve, circuit):
    """Host half of AHP::index (indexer.rs:70-96): synthesis, make_matrices_square, balance_matrices, per-row column
    sort.  -> (constraint system, [a, b, c]) with rows as lists of (coeff, column)."""
    c = get_curve(curve)
    cs = MarlinCS(c, assign=False)
    circuit.generate_constraints(cs)
    cs.make_matrices_square()
    mats = [[[(cf, j if kind == INPUT else cs.num_inputs + j) for cf, (kind, j) in row] for row in m]
            for m in (cs.at, cs.bt, cs.ct)]
    a, b, cc = mats
    da, db_ = sum(map(len, a)), sum(map(len, b))          # balance_matrices
    denser = da > db_
    for i in range(len(a)):
        if denser:
            la, lb = len(a[i]), len(b[i])
            a[i], b[i] = b[i], a[i]
            da += lb - la
            db_ += la - lb
            denser = da > db_
    for m in mats:
        for row in m:
            row.sort(key=lambda t: t[1])
    return cs, mats


def prepare_matrices(inst):
    """make_matrices_square + balance_matrices + per-row column sort on CSR index arrays (pure host code).
    -> (n rows = columns, padding variables added, [(row_ptr, col, coeff_mont, row_of_entry)] for A, B, C)"""
    nv, nc = inst.num_inputs + inst.num_aux, inst.num_constraints()
    pad_aux = max(nc - nv, 0)                   # make_matrices_square: dummy variables (value one) ...
    n = max(nv, nc)                             # ... or empty constraints
    mats = []
    for ptr, col, cf in (inst.csr("a"), inst.csr("b"), inst.csr("c")):
        ptr = np.asarray(ptr, dtype=np.int64)
        ptr = np.concatenate([ptr, np.full(n - nc, ptr[-1], dtype=np.int64)])
        mats.append((ptr, np.asarray(col, dtype=np.int64), np.asarray(cf, dtype=np.uint64).reshape(-1, 4)))
    # balance_matrices (constraint_systems.rs): greedy row swaps while A is the denser matrix
    la, lb = np.diff(mats[0][0]).tolist(), np.diff(mats[1][0]).tolist()
    da, db_ = sum(la), sum(lb)
    swap = np.zeros(n, dtype=bool)
    denser = da > db_
    for i in range(n):
        if not denser:
            break
        swap[i] = True
        da += lb[i] - la[i]
        db_ += la[i] - lb[i]
        denser = da > db_

    def select(mask, P, Q):
        lp, lq = np.diff(P[0]), np.diff(Q[0])
        ln = np.where(mask, lp, lq)
        ptr = np.concatenate([[0], np.cumsum(ln)]).astype(np.int64)
        start = np.where(mask, P[0][:-1], Q[0][:-1] + len(P[1]))
        src = np.repeat(start, ln) + (np.arange(ptr[-1]) - np.repeat(ptr[:-1], ln))
        return ptr, np.concatenate([P[1], Q[1]])[src], np.concatenate([P[2], Q[2]])[src]

    if swap.any():
        mats[0], mats[1] = select(swap, mats[1], mats[0]), select(swap, mats[0], mats[1])
    sorted_mats = []
    for ptr, col, cf in mats:
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(ptr))
        order = np.lexsort((col, rows))         # stable: by row, then by column (`row.sort_by_key(col)`)
        sorted_mats.append((ptr, col[order], cf[order], rows))
    return n, pad_aux, sorted_mats


def index_verifier_key(didx, ck: kzg10.CommitterKey, index_comms: dict, vk_g2) -> dict:
    """IndexVerifierKey (marlin/src/data_structures.rs:10-15) in the shape the transcript serialises (lib.rs:79-84):
    index_info, the 12 index commitments in `Index::iter` order, VerifierKey {g, gamma_g, h, beta_h, supported_degree}.
    vk_g2 = (h, beta_h): the G2 half of the SRS (never used by the prover except inside the transcript seed)."""
    c = didx.curve
    first = lambda q: codec.g1_from_mont(q[0][:1], q[1][:1], c)[0]
    return dict(num_variables=didx.nrows, num_constraints=didx.nrows, num_non_zeros=didx.num_non_zeros,
                index_comms=[index_comms[l] for l in INDEX_LABELS], g=first(ck.host_g), gamma_g=first(ck.host_gamma_g),
                h=vk_g2[0], beta_h=vk_g2[1], supported_degree=didx.max_degree)


class NativeIndex:
    """zkp_marlin_index: the arithmetization resident in HBM, built by the library from the three square CSR matrices
    (the index-manipulation half of AHP::index — make_matrices_square, balance_matrices, column sort — is `prepare_matrices`)."""

    def __init__(self, ctx: Context, inst):
        self.ctx, self.curve = ctx, get_curve(inst.curve)
        n, pad_aux, mats = prepare_matrices(inst)
        d = _lib.MarlinIndexDesc()
        d.curve, d.num_inputs, d.n, d.pad_aux = self.curve.cid, inst.num_inputs, n, pad_aux
        keep = []

        def P(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            return a.ctypes.data

        for name, (ptr, col, cf, _rows) in zip("abc", mats):
            m = getattr(d, name)
            m.row_ptr, m.col = P(ptr, np.uint32), P(col if len(col) else np.zeros(1), np.uint32)
            m.coeff = P(cf if len(cf) else np.zeros((1, 4)), np.uint64)
        h = C.c_void_p()
        _lib.check(ctx.lib.zkp_marlin_index_upload(ctx.h, C.byref(d), C.byref(h)), "zkp_marlin_index_upload")
        self.h = h
        info = (C.c_uint64 * 6)()
        _lib.check(ctx.lib.zkp_marlin_index_info(self.h, info), "zkp_marlin_index_info")
        self.xs, self.hs, self.ks, self.bs, self.max_degree, self.num_non_zeros = (int(v) for v in info)
        self.nrows, self.pad_aux, self.num_inputs = n, pad_aux, inst.num_inputs

    def commit_index(self, ck: kzg10.CommitterKey) -> dict:
        xy = np.zeros((12, 12), dtype=np.uint64)
        inf = np.zeros(12, dtype=np.uint8)
        _lib.check(self.ctx.lib.zkp_marlin_index_commit(self.ctx.h, self.h, ck.powers_of_g.handle, api._ptr(xy), api._ptr(inf)),
                   "zkp_marlin_index_commit")
        w = 2 * self.curve.fq_limbs
        pts = codec.g1_from_mont(np.ascontiguousarray(xy[:, :w]), inf, self.curve)
        return {l: (p, None) for l, p in zip(INDEX_LABELS, pts)}

    def free(self):
        if self.h:
            _lib.check(self.ctx.lib.zkp_marlin_index_free(self.ctx.h, self.h), "zkp_marlin_index_free")
            self.h = None


def prove_native(ctx: Context, nidx: NativeIndex, ck: kzg10.CommitterKey, ivk: dict | None, x, w_mont, rnd, ch: dict | None = None):
    """zkp_marlin_prove.  ivk: index verifier key dict (its to_bytes seeds the transcript) — create_random_proof; or
    ch: dict of fixed verifier messages (test hook).  x: formatted inputs (ints, leading one included); w_mont: witness
    (n_w, 4) Montgomery or list of ints.  Returns the same dictionary shape as `create_proof`."""
    from .fs_rng import index_verifier_key_bytes
    c = nidx.curve
    mont = lambda v: np.ascontiguousarray(codec.fr_to_mont(list(v), c).reshape(-1, 4))
    xm = mont(x)
    wm = np.ascontiguousarray(w_mont if isinstance(w_mont, np.ndarray) else mont(w_mont))
    keep = [xm, wm]
    R = _lib.MarlinRand()

    def P(a):
        keep.append(a)
        return a.ctypes.data

    R.w, R.z_a, R.z_b = P(mont(rnd["w"])), P(mont(rnd["z_a"])), P(mont(rnd["z_b"]))
    if rnd.get("mask_dev"):                                  # the mask polynomial already resident in HBM (3|H| Fr)
        R.mask, R.mask_on_device = rnd["mask_dev"], 1
    else:
        mask = rnd["mask"] if isinstance(rnd["mask"], np.ndarray) else mont(rnd["mask"])
        R.mask, R.mask_on_device = P(np.ascontiguousarray(mask)), 0
    R.blind_w, R.blind_z_a, R.blind_z_b = (P(mont(rnd["blind"][l])) for l in ("w", "z_a", "z_b"))
    R.blind_g_1, R.blind_shifted_g_1 = P(mont(rnd["blind"]["g_1"])), P(mont(rnd["blind_shifted"]["g_1"]))
    fixed = None
    ivk_b = b""
    if ch is not None:
        fixed = mont([ch[k] for k in ("alpha", "eta_a", "eta_b", "eta_c", "beta", "gamma", "xi")])
        keep.append(fixed)
    else:
        ivk_b = index_verifier_key_bytes(ivk, c)
    ivk_buf = (C.c_uint8 * max(len(ivk_b), 1)).from_buffer_copy(ivk_b or b"\x00")
    out = _lib.MarlinProof()
    _lib.check(ctx.lib.zkp_marlin_prove(ctx.h, nidx.h, ck.powers_of_g.handle, ck.powers_of_gamma_g.handle,
                                        C.cast(ivk_buf, C.c_void_p) if ch is None else None, len(ivk_b), api._ptr(xm),
                                        api._ptr(wm), wm.shape[0], C.byref(R), None if fixed is None else api._ptr(fixed),
                                        C.byref(out)), "zkp_marlin_prove")
    w2 = 2 * c.fq_limbs
    g1 = lambda words, inf: codec.g1_from_mont(np.array(words[:w2], dtype=np.uint64).reshape(1, w2), [inf], c)[0]
    labels = LABELS_1 + LABELS_2 + LABELS_3
    comm = np.array(out.comm, dtype=np.uint64).reshape(9, 12)
    sh = np.array(out.shifted, dtype=np.uint64).reshape(2, 12)
    comms = {}
    for i, l in enumerate(labels):
        s = None
        if l == "g_1":
            s = g1(sh[0], out.shifted_inf[0])
        elif l == "g_2":
            s = g1(sh[1], out.shifted_inf[1])
        comms[l] = (g1(comm[i], out.comm_inf[i]), s)
    chs = codec.fr_from_mont(np.array(out.challenges, dtype=np.uint64).reshape(7, 4), c)
    chd = dict(zip(("alpha", "eta_a", "eta_b", "eta_c", "beta", "gamma", "xi"), chs))
    evals = codec.fr_from_mont(np.array(out.evaluations, dtype=np.uint64).reshape(-1, 4), c)
    query = sorted([(l, chd["beta"]) for l in LABELS_1 + LABELS_2] + [(l, chd["gamma"]) for l in LABELS_3 + INDEX_LABELS])
    ow = np.array(out.opening_w, dtype=np.uint64).reshape(2, 12)
    rv = codec.fr_from_mont(np.array(out.opening_rand_v, dtype=np.uint64).reshape(2, 4), c)
    proofs = [(g1(ow[k], out.opening_w_inf[k]), rv[k] if out.opening_has_rand[k] else None)
              for k in range(out.num_opening_proofs)]
    tm = _lib.MarlinTiming()
    _lib.check(ctx.lib.zkp_marlin_last_timing(ctx.h, C.byref(tm)), "zkp_marlin_last_timing")
    timing = dict(ms_round=[float(x) for x in tm.ms_round], ms_commit=[float(x) for x in tm.ms_commit],
                  ms_evaluations=float(tm.ms_evaluations), ms_open=float(tm.ms_open), ms_total=float(tm.ms_total),
                  commit_points=int(tm.commit_points), open_points=int(tm.open_points), ntt_count=int(tm.ntt_count),
                  ntt_elements=int(tm.ntt_elements))
    return dict(commitments=comms, evaluations=evals, opening_proofs=proofs, query=query, challenges=chd, timing=timing)


def create_random_proof(ctx: Context, nidx: NativeIndex, ck: kzg10.CommitterKey, ivk: dict, circuit, rnd) -> dict:
    """marlin::create_random_proof (lib.rs:97-181).  circuit: a synthesizer (its assignment is taken through `MarlinCS`) or
    (formatted inputs incl. the leading one, witness); rnd: the zk randomness the reference draws from `zk_rng`."""
    if hasattr(circuit, "generate_constraints"):
        cs = MarlinCS(nidx.curve, assign=True)
        circuit.generate_constraints(cs)
        x, w = cs.input_assignment, cs.aux_assignment
    else:
        x, w = circuit
    return prove_native(ctx, nidx, ck, ivk, list(x), w, rnd)
