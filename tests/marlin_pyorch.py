"""TEST INFRASTRUCTURE (moved out of the product package in round 4): the Python-orchestrated device-resident Marlin prover —
the algorithm of marlin/src/lib.rs:97-181, ahp/prover.rs:86-427, pc/mod.rs:34-160 with every vector in HBM and every step a
C-ABI call on device pointers (NTTs, element-wise arithmetic, batch inversions, sparse products, gathers, vanishing-polynomial
folds, evaluations, witness division, KZG10 MSMs).  The product prover is csrc/marlin.hip behind zkp_marlin_prove
(ckb_zkp_amd/marlin.py); this one cross-checks it round by round and exercises the vector / polynomial entry points of the ABI.

`DeviceIndex` is the prover-side image of `Index` / `IndexProverKey` (ahp/indexer.rs:36-68): matrices as CSR, their
reindexed transposes for `t`, the arithmetization polynomials and their evaluations on K and B.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ckb_zkp_amd import _lib, api, codec, kzg10
from ckb_zkp_amd.api import Context
from ckb_zkp_amd.marlin import (INDEX_LABELS, LABELS_1, LABELS_2, LABELS_3, ZK_BOUND, MarlinCS, NativeIndex, _next_pow2,  # noqa: F401
                                index_matrices, index_verifier_key, prepare_matrices, prove_native, reindex_by_subdomain)
from ckb_zkp_amd.params import get_curve


class DVec:
    """n Fr elements (Montgomery) in device memory.  Slices are non-owning views."""

    def __init__(self, be, ptr: int, n: int, owner: bool = True):
        self.be, self.ptr, self.n, self.owner = be, ptr, n, owner

    def __len__(self):
        return self.n

    def view(self, a: int, b: int | None = None) -> "DVec":
        b = self.n if b is None else b
        assert 0 <= a <= b <= self.n
        return DVec(self.be, self.ptr + 32 * a, b - a, owner=False)


class DeviceBackend:
    def __init__(self, ctx: Context, curve):
        self.ctx, self.c = ctx, get_curve(curve)
        self.lib, self.h, self.cid = ctx.lib, ctx.h, self.c.cid
        self._live = []
        # size-keyed free lists shared by every backend of this context: a proof allocates the same ~120 vectors every
        # time, so after the first proof no hipMalloc / hipFree (each a device-wide synchronisation) is left in the path
        if not hasattr(ctx, "_fr_pool"):
            ctx._fr_pool = {}
        self._pool = ctx._fr_pool

    # ---- memory
    def alloc(self, n: int) -> DVec:
        nbytes = max(n, 1) * 32
        free = self._pool.get(nbytes)
        ptr = free.pop() if free else self.ctx.dev_alloc(nbytes)
        v = DVec(self, ptr, n)
        v.nbytes = nbytes
        self._live.append(v)
        return v

    def zeros(self, n: int) -> DVec:
        v = self.alloc(n)
        _lib.check(self.lib.zkp_dev_zero(self.h, C.c_void_p(v.ptr), n * 32), "zkp_dev_zero")
        return v

    def upload(self, ints) -> DVec:
        a = codec.fr_to_mont(ints, self.c).reshape(-1, 4)
        v = self.alloc(len(ints))
        if len(ints):
            self.ctx.h2d(v.ptr, a)
        return v

    def upload_mont(self, a: np.ndarray) -> DVec:
        v = self.alloc(a.shape[0])
        if a.shape[0]:
            self.ctx.h2d(v.ptr, a)
        return v

    def upload_raw(self, a: np.ndarray) -> int:
        p = self.ctx.to_device(np.ascontiguousarray(a))
        v = DVec(self, p, 0)
        v.nbytes = None                                      # not pooled
        self._live.append(v)
        return p

    def download(self, v: DVec) -> list:
        out = np.zeros((v.n, 4), dtype=np.uint64)
        if v.n:
            self.ctx.d2h(out, v.ptr)
        return codec.fr_from_mont(out, self.c)

    def copy_into(self, dst: DVec, src: DVec):
        assert dst.n >= src.n
        _lib.check(self.lib.zkp_d2d(self.h, C.c_void_p(dst.ptr), C.c_void_p(src.ptr), src.n * 32), "zkp_d2d")

    def pad(self, v: DVec, n: int) -> DVec:
        """copy of v zero-extended (or truncated) to n elements"""
        out = self.zeros(n)
        self.copy_into(out, v.view(0, min(v.n, n)))
        return out

    def shift(self, v: DVec, s: int) -> DVec:
        out = self.zeros(v.n + s)
        self.copy_into(out.view(s), v)
        return out

    def release_all(self):
        """hand every vector of this backend back to the context's pool (stream-ordered reuse: all work is enqueued on
        the context's stream, so the next user of a buffer runs after its last reader)"""
        for v in self._live:
            if v.owner and v.ptr:
                if getattr(v, "nbytes", None):
                    self._pool.setdefault(v.nbytes, []).append(v.ptr)
                else:
                    self.ctx.dev_free(v.ptr)
                v.ptr = 0
        self._live = []

    @staticmethod
    def trim_pool(ctx: Context):
        for lst in getattr(ctx, "_fr_pool", {}).values():
            for p in lst:
                ctx.dev_free(p)
        ctx._fr_pool = {}

    # ---- arithmetic
    def _k(self, k):
        return None if k is None else api._ptr(codec.fr_to_mont([k % self.c.r], self.c)[0])

    def _op(self, op, a: DVec, b: DVec | None, out: DVec, n: int, k=None):
        _lib.check(self.lib.zkp_fr_vec_op_dev(self.h, self.cid, op, C.c_void_p(a.ptr), C.c_void_p(b.ptr if b else 0),
                                              self._k(k), C.c_void_p(out.ptr), n), "zkp_fr_vec_op_dev")

    def mul(self, a: DVec, b: DVec) -> DVec:
        assert a.n == b.n
        out = self.alloc(a.n)
        self._op(api.VEC_MUL, a, b, out, a.n)
        return out

    def scale(self, a: DVec, k: int) -> DVec:
        out = self.alloc(a.n)
        self._op(api.VEC_SCALE, a, None, out, a.n, k)
        return out

    def addc(self, a: DVec, k: int) -> DVec:
        out = self.alloc(a.n)
        self._op(5, a, None, out, a.n, k)
        return out

    def axpy(self, a: DVec, b: DVec, k: int) -> DVec:
        """a + k*b with zero extension to max(len)"""
        n = max(a.n, b.n)
        out = self.pad(a, n)
        self._op(api.VEC_AXPY, out.view(0, b.n), b, out.view(0, b.n), b.n, k)
        return out

    def axpy_into(self, acc: DVec, b: DVec, k: int, at: int = 0):
        """acc[at : at + len(b)] += k * b"""
        tgt = acc.view(at, at + b.n)
        self._op(api.VEC_AXPY, tgt, b, tgt, b.n, k)

    def add_at(self, v: DVec, i: int, k: int):
        e = v.view(i, i + 1)
        self._op(5, e, None, e, 1, k)

    def sub(self, a: DVec, b: DVec) -> DVec:
        return self.axpy(a, b, -1)

    def binv(self, a: DVec) -> DVec:
        out = self.pad(a, a.n)
        _lib.check(self.lib.zkp_fr_batch_inverse_dev(self.h, self.cid, C.c_void_p(out.ptr), out.n), "zkp_fr_batch_inverse_dev")
        return out

    def ntt(self, v: DVec, size: int, op: int) -> DVec:
        out = self.pad(v, size)
        self.ctx.ntt_dev(self.c, out.ptr, size.bit_length() - 1, op)
        return out

    def fft(self, v, size):
        return self.ntt(v, size, api.NTT_FFT)

    def ifft(self, v, size=None):
        return self.ntt(v, size or _next_pow2(v.n), api.NTT_IFFT)

    def pmul(self, a: DVec, b: DVec) -> DVec:
        size = _next_pow2(a.n + b.n - 1)
        return self.ifft(self.mul(self.fft(a, size), self.fft(b, size))).view(0, a.n + b.n - 1)

    def evaluate(self, v: DVec, z: int) -> int:
        if v.n == 0:
            return 0
        out = self.ctx.poly_evaluate(self.c, v.ptr, v.n, codec.fr_to_mont([z], self.c)[0])
        return codec.fr_from_mont(out.reshape(1, 4), self.c)[0]

    def fold(self, v: DVec, n: int):
        """divide_by_vanishing_poly: -> (q [len - n], rem [n])"""
        q = self.alloc(max(v.n - n, 0))
        rem = self.alloc(n)
        _lib.check(self.lib.zkp_poly_divide_by_vanishing_dev(self.h, self.cid, C.c_void_p(v.ptr), v.n, n,
                                                             C.c_void_p(q.ptr if q.n else 0), C.c_void_p(rem.ptr)),
                   "zkp_poly_divide_by_vanishing_dev")
        return q, rem

    def element(self, v: DVec, i: int) -> int:
        return self.download(v.view(i, i + 1))[0]

    def spmv(self, csr, x: DVec, nrows: int) -> DVec:
        out = self.alloc(nrows)
        _lib.check(self.lib.zkp_fr_spmv_dev(self.h, self.cid, C.c_void_p(csr[0]), C.c_void_p(csr[1]), C.c_void_p(csr[2]),
                                            nrows, C.c_void_p(x.ptr), C.c_void_p(out.ptr)), "zkp_fr_spmv_dev")
        return out

    def gather(self, v: DVec, idx_dev: int, n: int) -> DVec:
        out = self.alloc(n)
        _lib.check(self.lib.zkp_fr_gather_dev(self.h, C.c_void_p(v.ptr), C.c_void_p(idx_dev), n, C.c_void_p(out.ptr)),
                   "zkp_fr_gather_dev")
        return out

    def msm(self, bases, v: DVec, offset: int = 0) -> np.ndarray:
        return bases.msm_mont_dev(v.ptr, v.n, offset=offset)


def _csr_dev(be: DeviceBackend, rows, ncols_unused=None):
    """list of rows [(coeff, col)] -> device (row_ptr, col, coeff) pointers"""
    rp = np.zeros(len(rows) + 1, dtype=np.uint32)
    cols, cfs = [], []
    for i, row in enumerate(rows):
        for cf, j in row:
            cols.append(j)
            cfs.append(cf)
        rp[i + 1] = len(cols)
    cf = codec.fr_to_mont(cfs, be.c).reshape(-1, 4) if cfs else np.zeros((1, 4), dtype=np.uint64)
    cl = np.asarray(cols if cols else [0], dtype=np.uint32)
    return (be.upload_raw(rp), be.upload_raw(cl), be.upload_raw(cf))


def _np_reindex(hs: int, xs: int, j: np.ndarray) -> np.ndarray:
    """vectorised reindex_by_subdomain (ahp/constraint_systems.rs `reindex_by_subdomain`)"""
    period = hs // xs
    j = j.astype(np.int64)
    i = j - xs
    return np.where(j < xs, j * period, i + i // max(period - 1, 1) + 1)


def _csr_upload(be: DeviceBackend, ptr, col, cf_mont):
    cf = cf_mont if len(cf_mont) else np.zeros((1, 4), dtype=np.uint64)
    cl = col if len(col) else np.zeros(1, dtype=np.uint32)
    return (be.upload_raw(np.asarray(ptr, dtype=np.uint32)), be.upload_raw(np.asarray(cl, dtype=np.uint32)),
            be.upload_raw(np.ascontiguousarray(cf, dtype=np.uint64)))


class DeviceIndex:
    """Prover-side index resident in HBM.  Built either from a host index (`marlin.index`, small circuits / tests) or
    directly from array-form matrices (`from_instance`), in which case the arithmetization itself — row/col/val over K,
    their interpolations and the evaluations over B (arithmetic.rs:98-172) — is computed on the device."""

    def __init__(self, ctx: Context, curve):
        self.curve = get_curve(curve)
        self.be = DeviceBackend(ctx, self.curve)

    def _finish(self, xs, hs, ks, bs, nrows):
        self.xs, self.hs, self.ks, self.bs, self.nrows = xs, hs, ks, bs, nrows
        self.max_degree = max(3 * hs + 2 * ZK_BOUND - 1, 3 * ks - 3)
        ratio = hs // xs
        i = np.arange(hs, dtype=np.int64)
        # w_evals_on_h[i] = 0 if i % ratio == 0 else w_ext[i - i/ratio - 1] - x_evals_on_h[i]   (prover.rs:176-186)
        self.w_idx = self.be.upload_raw(np.where(i % ratio == 0, -1, i - i // ratio - 1).astype(np.int32))
        self.x_idx = self.be.upload_raw(np.where(i % ratio == 0, -1, i).astype(np.int32))
        self.be.ctx.sync()
        # index data stays resident for the lifetime of the context; everything else the construction allocated goes back
        keep = {self.h_el.ptr, self.w_idx, self.x_idx}
        for m in "abc":
            keep.update(self.csr[m])
            keep.update(self.csr_t[m])
            keep.update(v.ptr for v in self.on_k[m].values())
            keep.update(v.ptr for v in self.on_b[m].values())
        keep.update(v.ptr for v in self.polys.values())
        for v in self.be._live:
            if v.owner and v.ptr and v.ptr not in keep:
                self.be.ctx.dev_free(v.ptr)
                v.ptr = 0
        self.be._live = []
        self._owned = keep

    @classmethod
    def from_host_index(cls, ctx: Context, idx):
        self = cls(ctx, idx["curve"])
        be = self.be
        hs, xs = idx["hs"], idx["xs"]
        self.csr = {m: _csr_dev(be, idx[m]) for m in "abc"}
        # transposed, re-indexed matrices: t_on_h[k] = sum_{(i, j): reindex(j) = k} coeff * r_alpha[i]  (prover.rs:259-269)
        self.csr_t = {}
        for m in "abc":
            rows = [[] for _ in range(hs)]
            for i, row in enumerate(idx[m]):
                for cf, j in row:
                    rows[reindex_by_subdomain(hs, xs, j)].append((cf, i))
            self.csr_t[m] = _csr_dev(be, rows)
        self.h_el = be.upload(idx["h_el"])
        S = idx["star"]
        self.on_k = {m: {k: be.upload(S[m]["on_k"][k]) for k in ("row", "col", "val")} for m in "abc"}
        self.on_b = {m: {k: be.upload(S[m]["on_b"][k]) for k in ("row", "col", "val", "row_col")} for m in "abc"}
        self.polys = {f"{m}_{k}": be.upload(S[m]["polys"][k]) for m in "abc" for k in ("row", "col", "val", "row_col")}
        self.num_non_zeros = max(sum(len(row) for row in idx[m]) for m in "abc")
        self._finish(xs, hs, idx["ks"], idx["bs"], len(idx["a"]))
        return self

    @classmethod
    def from_instance(cls, ctx: Context, inst):
        """AHP::index (indexer.rs:70-117) on array-form matrices: make_matrices_square, balance_matrices, per-row
        column sort (prepare_matrices, host index arrays), then compose_matrix_polynomials on the device.
        inst: r1cs.R1csInstance (CSR, Montgomery coeffs)."""
        self = cls(ctx, inst.curve)
        be, c = self.be, self.curve
        ni = inst.num_inputs
        n, self.pad_aux, sorted_mats = prepare_matrices(inst)
        nnz = max(int(m[0][-1]) for m in sorted_mats)
        xs, hs, ks = _next_pow2(ni), _next_pow2(n), _next_pow2(nnz)
        bs = _next_pow2(3 * ks - 3)
        # H as a device vector: the evaluations of X over the domain
        xpoly = be.zeros(hs)
        be.add_at(xpoly, 1, 1)
        self.h_el = be.fft(xpoly, hs)
        diag_inv = be.scale(self.h_el, pow(hs, -1, c.r))          # 1 / (|H| u^-1) = u / |H|
        self.csr, self.csr_t, self.on_k, self.on_b, self.polys = {}, {}, {}, {}, {}
        for m, (ptr, col, cf, rows) in zip("abc", sorted_mats):
            self.csr[m] = _csr_upload(be, ptr, col, cf)
            jj = _np_reindex(hs, xs, col)
            order = np.argsort(jj, kind="stable")
            tptr = np.concatenate([[0], np.cumsum(np.bincount(jj, minlength=hs))])
            self.csr_t[m] = _csr_upload(be, tptr, rows[order], cf[order])
            k = len(col)
            pad0 = lambda a, fill: np.concatenate([a, np.full(ks - k, fill, dtype=np.int64)]).astype(np.int32)
            row = be.gather(self.h_el, be.upload_raw(pad0(jj, 0)), ks)
            colv = be.gather(self.h_el, be.upload_raw(pad0(rows, 0)), ks)
            vcf = be.upload_mont(np.concatenate([cf, np.zeros((ks - k, 4), dtype=np.uint64)]))
            val = be.mul(vcf, be.gather(diag_inv, be.upload_raw(pad0(jj, -1)), ks))
            rc = be.mul(row, colv)
            self.on_k[m] = dict(row=row, col=colv, val=val)
            self.on_b[m] = {}
            for name, ev in (("row", row), ("col", colv), ("val", val), ("row_col", rc)):
                p = be.ifft(ev, ks)
                self.polys[f"{m}_{name}"] = p
                self.on_b[m][name] = be.fft(p, bs)
        self.num_non_zeros = nnz
        self._finish(xs, hs, ks, bs, n)
        return self

    def free(self):
        """release the resident index"""
        for p in getattr(self, "_owned", ()):
            if p:
                self.be.ctx.dev_free(p)
        self._owned = set()

    def commit_index(self, ctx: Context, ck: kzg10.CommitterKey):
        """index commitments (the verifier key's half of AHP::index, lib.rs:69-86) -> {label: (affine, None)}"""
        out = {}
        for l, p in self.polys.items():
            xy, inf = ctx.into_affine(self.curve, 1, ck.powers_of_g.msm_mont_dev(p.ptr, p.n))
            out[l] = (codec.g1_from_mont(xy, [inf], self.curve)[0], None)
        return out


def create_random_proof(ctx: Context, didx: DeviceIndex, ck: kzg10.CommitterKey, ivk: dict, circuit, rnd,
                        timing: dict | None = None):
    """marlin::create_random_proof (lib.rs:97-181): the verifier messages are DERIVED from the Fiat–Shamir transcript
    (library FiatShamirRng seeded with to_bytes![ivk, public_input]).  circuit: a synthesizer or (formatted inputs incl.
    the leading one, witness)."""
    from ckb_zkp_amd.fs_rng import FiatShamirChallenger
    if hasattr(circuit, "generate_constraints"):
        cs = MarlinCS(didx.curve, assign=True)
        circuit.generate_constraints(cs)
        public = cs.input_assignment[1:]
    else:
        public = list(circuit[0])[1:]
    chal = FiatShamirChallenger(didx.curve, didx.hs, ivk, public)
    return create_proof(ctx, didx, ck, circuit, rnd, chal, timing)


def create_proof(ctx: Context, didx: DeviceIndex, ck: kzg10.CommitterKey, circuit, rnd, ch, timing: dict | None = None):
    """Device-resident Marlin prover, round by round as lib.rs:97-181 prescribes: AHP round -> PC::commit of the round's
    oracles (one batched MSM call) -> absorb -> next verifier message.  ch: a challenger (fs_rng.FiatShamirChallenger via
    `create_random_proof`; fs_rng.FixedChallenger or a plain dict = test hook with supplied messages).  rnd: the zk
    randomness (masks, blinders) as in marlin.create_proof.  Returns commitments, evaluations (query order), opening
    proofs and the challenges used, as canonical integers."""
    import time
    from ckb_zkp_amd.fs_rng import FixedChallenger
    chal = FixedChallenger(ch) if isinstance(ch, dict) else ch
    c = didx.curve
    r = c.r
    be = DeviceBackend(ctx, c)
    xs, hs, ks, bs = didx.xs, didx.hs, didx.ks, didx.bs
    D = didx.max_degree
    t_start = time.perf_counter()
    try:
        # ---- prover_init: synthesis on the host, everything else on the device
        if hasattr(circuit, "generate_constraints"):
            cs = MarlinCS(c, assign=True)
            circuit.generate_constraints(cs)
            cs.make_matrices_square()
            x, w = cs.input_assignment, cs.aux_assignment
            z = be.upload(x + w)
        else:                                   # (formatted inputs, witness) as canonical integers or Montgomery rows
            x, w = circuit
            x = list(x)
            wm = w if isinstance(w, np.ndarray) else codec.fr_to_mont(list(w), c).reshape(-1, 4)
            pad = codec.fr_to_mont([1] * getattr(didx, "pad_aux", 0), c).reshape(-1, 4)
            z = be.upload_mont(np.concatenate([codec.fr_to_mont(x, c).reshape(-1, 4), wm, pad]))
        ctx.sync()
        t0 = time.perf_counter()
        z_a_ev = be.spmv(didx.csr["a"], z, didx.nrows)
        z_b_ev = be.spmv(didx.csr["b"], z, didx.nrows)
        # ---- first round (prover.rs:150-222)
        x_poly = be.ifft(be.upload(x), xs)
        x_on_h = be.fft(x_poly, hs)
        w_ext = be.pad(z.view(len(x)), hs - xs)
        w_on_h = be.sub(be.gather(w_ext, didx.w_idx, hs), be.gather(x_on_h, didx.x_idx, hs))

        def masked(ev: DVec, rand_coeff: int) -> DVec:      # interpolate(ev) + rand * v_H
            p = be.pad(be.ifft(ev, hs), hs + 1)
            be.add_at(p, 0, -rand_coeff)
            be.add_at(p, hs, rand_coeff)
            return p

        w_poly, _ = be.fold(masked(w_on_h, rnd["w"][0]), xs)
        z_a, z_b = masked(z_a_ev, rnd["z_a"][0]), masked(z_b_ev, rnd["z_b"][0])
        mask = be.upload_mont(rnd["mask"]) if isinstance(rnd["mask"], np.ndarray) else be.upload(rnd["mask"])
        _, mrem = be.fold(mask, hs)
        be.add_at(mask, 0, -be.element(mrem, 0))
        polys = dict(didx.polys)
        polys.update(w=w_poly, z_a=z_a, z_b=z_b, mask=mask)
        # ---- PC::commit (pc/mod.rs:34-71) of one round's oracles: MSMs against the resident powers, one batched call per
        # base vector (the MSMs of the round overlap on the context's MSM streams)
        bounds = {"g_1": hs - 2, "g_2": ks - 2}
        hide = lambda l: l in ("w", "z_a", "z_b", "g_1")
        blind_dev = {l: be.upload(rnd["blind"][l]) for l in ("w", "z_a", "z_b", "g_1")}
        blind_s_dev = {"g_1": be.upload(rnd["blind_shifted"]["g_1"])}
        comms = {}
        t_commit_acc = [0.0]

        def to_affine(jac):
            xy, inf = ctx.into_affine(c, 1, jac)
            return codec.g1_from_mont(xy, [inf], c)[0]

        def commit_round(labels):
            tc0 = time.perf_counter()
            jobs, slot = [], {}
            for l in labels:
                slot[(l, False)] = len(jobs)
                jobs.append((polys[l].ptr, polys[l].n, 0))
                if l in bounds:                                  # shifted_powers(bound) = powers[D - bound ..]
                    slot[(l, True)] = len(jobs)
                    jobs.append((polys[l].ptr, polys[l].n, D - bounds[l]))
            jac = ck.powers_of_g.msm_mont_batch_dev(jobs)
            bjobs, bslot = [], {}
            for l in labels:
                if hide(l):
                    bslot[(l, False)] = len(bjobs)
                    bjobs.append((blind_dev[l].ptr, 2, 0))
                    if l in bounds:
                        bslot[(l, True)] = len(bjobs)
                        bjobs.append((blind_s_dev[l].ptr, 2, 0))
            bjac = ck.powers_of_gamma_g.msm_mont_batch_dev(bjobs) if bjobs else []

            def point(l, shifted):
                j = jac[slot[(l, shifted)]]
                if (l, shifted) in bslot:
                    j = ctx.fold(c, 1, np.concatenate([j, bjac[bslot[(l, shifted)]]]))
                return to_affine(j)

            for l in labels:
                comms[l] = (point(l, False), point(l, True) if l in bounds else None)
            t_commit_acc[0] += time.perf_counter() - tc0
            return [comms[l] for l in labels]

        alpha, ea, eb, ec = chal.first(commit_round(LABELS_1))          # lib.rs:109-114
        # ---- second round (prover.rs:230-321)
        m_poly = be.axpy(be.axpy(be.scale(be.pmul(z_a, z_b), ec), z_a, ea), z_b, eb)
        v_alpha = (pow(alpha, hs, r) - 1) % r
        r_alpha_on_h = be.scale(be.binv(be.addc(be.scale(didx.h_el, r - 1), alpha)), v_alpha)
        r_alpha = be.ifft(r_alpha_on_h, hs)
        t_on_h = be.zeros(hs)
        for m, eta in (("a", ea), ("b", eb), ("c", ec)):
            be.axpy_into(t_on_h, be.spmv(didx.csr_t[m], r_alpha_on_h, hs), eta)
        t_poly = be.ifft(t_on_h, hs)
        z_poly = be.sub(be.shift(w_poly, xs), w_poly)          # w * v_X
        z_poly = be.axpy(z_poly, x_poly, 1)
        size = _next_pow2(max(mask.n, r_alpha.n + m_poly.n, t_poly.n + z_poly.n))
        prod = be.sub(be.mul(be.fft(r_alpha, size), be.fft(m_poly, size)), be.mul(be.fft(t_poly, size), be.fft(z_poly, size)))
        q1 = be.axpy(be.ifft(prod, size), mask, 1)
        h1, xg1 = be.fold(q1, hs)
        polys.update(t=t_poly, g_1=xg1.view(1, hs), h_1=h1.view(0, 2 * hs))
        beta = chal.second(commit_round(LABELS_2))                      # lib.rs:117-121
        # ---- third round (prover.rs:331-427)
        va, vb = v_alpha, (pow(beta, hs, r) - 1) % r
        acc = be.zeros(ks)
        for m, eta in (("a", ea), ("b", eb), ("c", ec)):
            ok = didx.on_k[m]
            inv = be.binv(be.mul(be.addc(be.scale(ok["row"], r - 1), beta), be.addc(be.scale(ok["col"], r - 1), alpha)))
            be.axpy_into(acc, be.mul(ok["val"], inv), eta)
        t3 = be.ifft(be.scale(acc, va * vb % r), ks)
        den = {}
        for m in "abc":
            ob = didx.on_b[m]
            d = be.axpy(be.axpy(ob["row_col"], ob["row"], -alpha), ob["col"], -beta)
            den[m] = be.addc(d, alpha * beta % r)
        a_on_b = be.zeros(bs)
        for m, eta, o1_, o2_ in (("a", ea, "b", "c"), ("b", eb, "c", "a"), ("c", ec, "a", "b")):
            be.axpy_into(a_on_b, be.mul(be.mul(didx.on_b[m]["val"], den[o1_]), den[o2_]), eta)
        a_poly = be.ifft(be.scale(a_on_b, va * vb % r), bs)
        b_poly = be.ifft(be.mul(be.mul(den["a"], den["b"]), den["c"]), bs)
        # degrees: a, b <= 3|K| - 3; t3 < |K|
        h2, _ = be.fold(be.sub(a_poly.view(0, 3 * ks - 2), be.pmul(b_poly.view(0, 3 * ks - 2), t3)), ks)
        polys.update(g_2=t3.view(1, ks), h_2=h2.view(0, 3 * ks - 3))
        gamma = chal.third(commit_round(LABELS_3))                      # lib.rs:124-128
        ctx.sync()
        t_rounds = time.perf_counter()
        t_commit = t_rounds
        # ---- evaluations + batch_open (lib.rs:147-165, pc/mod.rs:73-160)
        query = sorted([(l, beta) for l in LABELS_1 + LABELS_2] + [(l, gamma) for l in LABELS_3 + INDEX_LABELS])
        evals = [be.evaluate(polys[l], pt) for l, pt in query]
        xi = chal.opening(evals)                                        # lib.rs:157-158
        points = sorted({pt for _, pt in query})
        wjobs, rbs = [], []
        for pt in points:
            p = be.zeros(D + 1)
            rb, chal = [0, 0], 1
            for l in sorted(l for l, q in query if q == pt):
                be.axpy_into(p, polys[l], chal)
                if hide(l):
                    rb = [(rb[i] + chal * rnd["blind"][l][i]) % r for i in range(2)]
                if l in bounds:
                    sc = chal * xi % r
                    be.axpy_into(p, polys[l], sc, at=D - bounds[l])
                    if hide(l):
                        rb = [(rb[i] + sc * rnd["blind_shifted"][l][i]) % r for i in range(2)]
                chal = chal * xi % r * xi % r
            q = be.alloc(D)
            ctx.poly_div_linear(c, p.ptr, D + 1, codec.fr_to_mont([pt], c)[0], q.ptr)
            wjobs.append((q.ptr, D, 0))
            rbs.append(rb)
        wjac = ck.powers_of_g.msm_mont_batch_dev(wjobs)          # both witness MSMs in flight together
        proofs = []
        for pt, w_jac, rb in zip(points, wjac, rbs):
            rand_v = None
            if any(rb):
                rbd = be.upload(rb)
                qb = be.alloc(1)
                ev = ctx.poly_div_linear(c, rbd.ptr, 2, codec.fr_to_mont([pt], c)[0], qb.ptr)
                w_jac = ctx.fold(c, 1, np.concatenate([w_jac, be.msm(ck.powers_of_gamma_g, qb)]))
                rand_v = codec.fr_from_mont(ev.reshape(1, 4), c)[0]
            proofs.append((to_affine(w_jac), rand_v))
        ctx.sync()
        t_open = time.perf_counter()
        if timing is not None:
            timing.update(synthesis_upload_s=t0 - t_start, rounds_s=t_rounds - t0 - t_commit_acc[0], commit_s=t_commit_acc[0],
                          eval_open_s=t_open - t_commit, total_s=t_open - t0)
        return dict(commitments=comms, evaluations=evals, opening_proofs=proofs, query=query,
                    challenges=dict(alpha=alpha, eta_a=ea, eta_b=eb, eta_c=ec, beta=beta, gamma=gamma, xi=xi))
    finally:
        be.release_all()


# ---------------------------------------------------------------------------------------------------------------------
# The same prover behind the C ABI (csrc/marlin.hip): zkp_marlin_index_upload / zkp_marlin_prove.  The Python
# orchestration above is the host mirror the C++ was ported from; `NativeIndex` / `prove_native` are what a Rust caller gets.
