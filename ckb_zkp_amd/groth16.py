"""Host-side mirror of the reference's Groth16 prover interface, running on the MI355X backend.

Mirrors /root/reference/groth16/src/lib.rs:39-91 (`Parameters`, `Proof`, `create_random_proof`) and
/root/reference/groth16/src/prover.rs:97-211 (`create_random_proof`, `create_proof`, `create_proof_no_zk`).
Setup (`generate_parameters`, /root/reference/groth16/src/generator.rs:135-286) is NOT part of the hot path;
it is provided with an explicit trapdoor so that synthetic keys of 2^20+ constraints can be produced on the
GPU (zkp_fixed_base_mul_*) for the parity tests and the benchmark.
"""
from __future__ import annotations

import ctypes as C
import secrets
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .api import Context, _ptr
from .codec import fr_canonical, fr_to_mont, g1_from_mont, g1_to_mont, g2_from_mont, g2_to_mont, ints_to_limbs
from .params import CurveParams, get_curve
from .r1cs import ConstraintSystem, PolynomialDegreeTooLarge, R1csInstance


@dataclass
class Proof:
    """groth16/src/lib.rs:52-57 — affine points as canonical integers; None = identity."""
    a: tuple
    b: tuple
    c: tuple


@dataclass
class Parameters:
    """groth16/src/lib.rs:79-91 in ABI layout (Montgomery limbs + identity flags)."""
    curve: CurveParams
    num_inputs: int
    num_aux: int
    num_constraints: int
    alpha_g1: np.ndarray
    beta_g1: np.ndarray
    beta_g2: np.ndarray
    gamma_g2: np.ndarray
    delta_g1: np.ndarray
    delta_g2: np.ndarray
    gamma_abc_g1: tuple            # (xy, inf)
    a_query: tuple
    b_g1_query: tuple
    b_g2_query: tuple
    h_query: tuple
    l_query: tuple
    toxic: dict = field(default_factory=dict)   # trapdoor exponents (test keys only)


def _domain_log(c: CurveParams, n: int) -> int:
    lg = max(n - 1, 0).bit_length()
    if lg > c.two_adicity:
        raise PolynomialDegreeTooLarge()      # r1cs_to_qap.rs:123-125
    return lg


def _as_instance(curve, circuit, assign: bool) -> R1csInstance:
    if isinstance(circuit, R1csInstance):
        return circuit
    cs = ConstraintSystem(curve, assign)
    circuit.generate_constraints(cs)
    return R1csInstance.from_cs(cs)


def qap_exponents(inst: R1csInstance, tau: int):
    """R1CStoQAP::instance_map_with_evaluation (r1cs_to_qap.rs:58-110): a_i(tau), b_i(tau), c_i(tau), Z(tau)."""
    c = inst.curve
    r = c.r
    nc, ni = inst.num_constraints(), inst.num_inputs
    lg = _domain_log(c, nc + (ni - 1) + 1)
    N = 1 << lg
    w = pow(pow(c.fr_generator, (r - 1) >> c.two_adicity, r), 1 << (c.two_adicity - lg), r)
    zt = (pow(tau, N, r) - 1) % r
    # Lagrange coefficients u_i = Z(tau) w^i / (N (tau - w^i)), batch inversion
    els, p = [], 1
    for _ in range(N):
        els.append(p)
        p = p * w % r
    if zt == 0:
        u = [1 if e == tau % r else 0 for e in els]
    else:
        den = [(tau - e) % r for e in els]
        pref, acc = [], 1
        for d in den:
            pref.append(acc)
            acc = acc * d % r
        inv = pow(acc, -1, r)
        zn = zt * pow(N, -1, r) % r
        u = [0] * N
        for i in range(N - 1, -1, -1):
            u[i] = zn * els[i] % r * (inv * pref[i] % r) % r
            inv = inv * den[i] % r
    nvars = (ni - 1) + inst.num_aux
    Ri = pow(1 << (64 * c.fr_limbs), -1, r)
    out = []
    for which in "abc":
        row_ptr, col, coeff = inst.csr(which)
        from .codec import limbs_to_ints
        cf = [x * Ri % r for x in limbs_to_ints(coeff)] if len(col) else []
        acc = [0] * (nvars + 1)
        rp = row_ptr.tolist()
        cl = col.tolist()
        for i in range(nc):
            ui = u[i]
            for k in range(rp[i], rp[i + 1]):
                acc[cl[k]] = (acc[cl[k]] + ui * cf[k]) % r
        out.append(acc)
    a, b, cc = out
    for i in range(ni):
        a[i] = (a[i] + u[nc + i]) % r
    return a, b, cc, zt, N


class LazyInts:
    """A vector of Fr elements held as Montgomery limbs ((n, 4) uint64) that turns into a list of canonical Python ints on first
    use as a sequence (trapdoor exponents of large synthetic keys: 2^24 big ints are only built when a test asks for them;
    `.mont` is the array for callers that want it as it is)."""

    def __init__(self, mont: np.ndarray, c: CurveParams):
        self.mont, self._c, self._ints = mont, c, None

    def _get(self):
        if self._ints is None:
            from .codec import fr_from_mont
            self._ints = fr_from_mont(self.mont, self._c)
        return self._ints

    def __len__(self):
        return self.mont.shape[0]

    def __iter__(self):
        return iter(self._get())

    def __getitem__(self, i):
        return self._get()[i]

    def __eq__(self, other):
        return list(self) == list(other)


KEYGEN_DEVICE_MIN_LOG = 15      # domains >= 2^15: the exponent vectors are computed with the library's Fr vector kernels


def _generate_parameters_dev(ctx: Context, c: CurveParams, inst: R1csInstance, alpha, beta, gamma, delta, tau, g1_k, g2_k, lg):
    """generate_parameters for LARGE synthetic instances: the same formulas as `qap_exponents` + the loops of
    `generate_parameters` (generator.rs:135-286: Lagrange coefficients at tau by batch inversion, a_i / b_i / c_i(tau) as transposed
    sparse products, l = (beta a + alpha b + c) / delta, h_i = Z(tau) tau^i / delta) evaluated with the Fr vector primitives of the C
    ABI (zkp_fr_vec_op_dev, zkp_fr_batch_inverse_dev, zkp_fr_spmv_dev) instead of Python big-int loops: 2.5 min -> seconds at 2^24.
    Setup is not the hot path; tests/test_gpu_groth16.py checks this path against the big-int one on the same instance."""
    r = c.r
    nc, ni = inst.num_constraints(), inst.num_inputs
    N = 1 << lg
    nv = ni + inst.num_aux
    w = pow(pow(c.fr_generator, (r - 1) >> c.two_adicity, r), 1 << (c.two_adicity - lg), r)
    zt = (pow(tau, N, r) - 1) % r
    assert zt != 0, "tau inside the domain"
    m1 = lambda x: fr_to_mont([x % r], c)[0]
    R_inv = pow(1 << (64 * c.fr_limbs), -1, r)
    bufs = []

    def alloc(n):
        p = ctx.dev_alloc(max(n, 1) * 32)
        bufs.append(p)
        return p

    def powers(n, base, first):
        """first * base^i, i < n (Montgomery), by doubling"""
        buf = alloc(n)
        ctx.h2d(buf, m1(first).reshape(1, 4))
        ln = 1
        while ln < n:
            m = min(ln, n - ln)
            ctx.fr_vec_op(c, 3, buf, None, buf + ln * 32, m, m1(pow(base, ln, r)))
            ln += m
        return buf

    def download(buf, n, k=1, off=0):
        """canonical limbs of k * v[off + i], i < n: scaling by k / R leaves the canonical representation in memory"""
        tmp = alloc(n)
        ctx.fr_vec_op(c, 3, buf + off * 32, None, tmp, n, m1(k * R_inv % r))
        out = np.zeros((n, c.fr_limbs), dtype=np.uint64)
        if n:
            ctx.d2h(out, tmp)
        return out

    def mont_host(buf, n):
        out = np.zeros((n, c.fr_limbs), dtype=np.uint64)
        if n:
            ctx.d2h(out, buf)
        return out

    try:
        els = powers(N, w, 1)
        u = alloc(N)                                                   # u_i = Z(tau) w^i / (N (tau - w^i))
        ctx.fr_vec_op(c, 3, els, None, u, N, m1(r - 1))
        ctx.fr_vec_op(c, 5, u, None, u, N, m1(tau))
        ctx.fr_batch_inverse(c, u, N)
        ctx.fr_vec_op(c, 0, u, els, u, N)
        ctx.fr_vec_op(c, 3, u, None, u, N, m1(zt * pow(N, -1, r) % r))
        rows_of = None
        vecs = []
        for which in "abc":
            row_ptr, col, coeff = inst.csr(which)
            row_ptr = np.asarray(row_ptr, dtype=np.int64)
            col = np.asarray(col, dtype=np.int64)
            rows_of = np.repeat(np.arange(nc, dtype=np.int64), np.diff(row_ptr))
            order = np.argsort(col, kind="stable")
            tptr = np.zeros(nv + 1, dtype=np.uint32)
            np.cumsum(np.bincount(col, minlength=nv), out=tptr[1:])
            d_ptr = ctx.to_device(tptr)
            d_col = ctx.to_device(np.ascontiguousarray(rows_of[order] if len(col) else np.zeros(1), dtype=np.uint32))
            d_cf = ctx.to_device(np.ascontiguousarray(np.asarray(coeff, dtype=np.uint64).reshape(-1, 4)[order] if len(col) else np.zeros((1, 4)), dtype=np.uint64))
            bufs.extend([d_ptr, d_col, d_cf])
            acc = alloc(nv)
            ctx.fr_spmv(c, d_ptr, d_col, d_cf, nv, u, acc)              # acc_j = sum_{k: col_k = j} coeff_k u_{row_k}
            vecs.append(acc)
        a, b, cc = vecs
        ctx.fr_vec_op(c, 1, a, u + nc * 32, a, ni)                     # a_i += u_{nc + i}, i < num_inputs (r1cs_to_qap.rs:83-85)
        t = alloc(nv)                                                  # beta a + alpha b + c
        ctx.fr_vec_op(c, 3, a, None, t, nv, m1(beta))
        ctx.fr_vec_op(c, 4, t, b, t, nv, m1(alpha))
        ctx.fr_vec_op(c, 1, t, cc, t, nv)
        gi, di = pow(gamma, -1, r), pow(delta, -1, r)
        lvec = alloc(nv)
        ctx.fr_vec_op(c, 3, t, None, lvec, nv, m1(di))
        hvec = powers(N - 1, tau, zt * di % r)
        g1_base, _ = g1_to_mont([c.g1], c)
        g2_base, _ = g2_to_mont([c.g2], c)
        mul1 = lambda sc: ctx.fixed_base_mul(c, 1, g1_base, sc)
        mul2 = lambda sc: ctx.fixed_base_mul(c, 2, g2_base, sc)
        singles1, _ = mul1(fr_canonical([k * g1_k % r for k in (alpha, beta, delta)], c))
        singles2, _ = mul2(fr_canonical([k * g2_k % r for k in (beta, gamma, delta)], c))
        b_can1 = download(b, nv, g1_k)
        params = Parameters(
            curve=c, num_inputs=ni, num_aux=inst.num_aux, num_constraints=nc,
            alpha_g1=singles1[0], beta_g1=singles1[1], delta_g1=singles1[2],
            beta_g2=singles2[0], gamma_g2=singles2[1], delta_g2=singles2[2],
            gamma_abc_g1=mul1(download(t, ni, gi * g1_k % r)), a_query=mul1(download(a, nv, g1_k)), b_g1_query=mul1(b_can1),
            b_g2_query=mul2(b_can1 if g1_k == g2_k else download(b, nv, g2_k)),
            h_query=mul1(download(hvec, N - 1, g1_k)), l_query=mul1(download(lvec, nv - ni, g1_k, off=ni)),
            toxic=dict(alpha=alpha, beta=beta, gamma=gamma, delta=delta, tau=tau, g1_k=g1_k, g2_k=g2_k,
                       a=LazyInts(mont_host(a, nv), c), b=LazyInts(mont_host(b, nv), c), c=LazyInts(mont_host(cc, nv), c),
                       l=LazyInts(mont_host(lvec, nv), c), h=LazyInts(mont_host(hvec, N - 1), c), zt=zt))
    finally:
        ctx.sync()
        for p_ in bufs:
            ctx.dev_free(p_)
    return params


def generate_parameters(ctx: Context, curve, circuit, alpha: int, beta: int, gamma: int, delta: int, tau: int,
                        g1_k: int = 1, g2_k: int = 1, keygen: str = "auto") -> Parameters:
    """generator.rs:135-286 with the toxic waste (and the generator multiples) as explicit inputs.
    keygen: "host" = Python big-int exponents, "device" = the library's Fr vector kernels (`_generate_parameters_dev`),
    "auto" = device from 2^KEYGEN_DEVICE_MIN_LOG constraints on."""
    c = get_curve(curve)
    r = c.r
    inst = _as_instance(c, circuit, assign=False)
    lg_dom = _domain_log(c, inst.num_constraints() + (inst.num_inputs - 1) + 1)
    if keygen == "device" or (keygen == "auto" and lg_dom >= KEYGEN_DEVICE_MIN_LOG and pow(tau, 1 << lg_dom, r) != 1):
        return _generate_parameters_dev(ctx, c, inst, alpha % r, beta % r, gamma % r, delta % r, tau % r, g1_k, g2_k, lg_dom)
    a, b, cc, zt, N = qap_exponents(inst, tau)
    ni = inst.num_inputs
    gi, di = pow(gamma, -1, r), pow(delta, -1, r)
    gamma_abc = [(beta * a[i] + alpha * b[i] + cc[i]) * gi % r for i in range(ni)]
    l = [(beta * x + alpha * y + z) * di % r for x, y, z in zip(a, b, cc)]
    h, p, zd = [], 1, zt * di % r
    for _ in range(N - 1):
        h.append(zd * p % r)
        p = p * tau % r
    g1_base, _ = g1_to_mont([c.g1], c)
    g2_base, _ = g2_to_mont([c.g2], c)

    def mul1(ks):
        return ctx.fixed_base_mul(c, 1, g1_base, fr_canonical([k * g1_k % r for k in ks], c))

    def mul2(ks):
        return ctx.fixed_base_mul(c, 2, g2_base, fr_canonical([k * g2_k % r for k in ks], c))

    singles1, _ = mul1([alpha, beta, delta])
    singles2, _ = mul2([beta, gamma, delta])
    return Parameters(
        curve=c, num_inputs=ni, num_aux=inst.num_aux, num_constraints=inst.num_constraints(),
        alpha_g1=singles1[0], beta_g1=singles1[1], delta_g1=singles1[2],
        beta_g2=singles2[0], gamma_g2=singles2[1], delta_g2=singles2[2],
        gamma_abc_g1=mul1(gamma_abc), a_query=mul1(a), b_g1_query=mul1(b), b_g2_query=mul2(b),
        h_query=mul1(h), l_query=mul1(l[ni:]),
        toxic=dict(alpha=alpha, beta=beta, gamma=gamma, delta=delta, tau=tau, g1_k=g1_k, g2_k=g2_k,
                   a=a, b=b, c=cc, l=l, h=h, zt=zt),
    )


def assemble(ctx: Context, curve, sums_xyz: np.ndarray, r: int, s: int):
    """zkp_groth16_assemble: folded partial sums (A|B1|B2|H|L Jacobian) -> (proof limbs, identity flags)."""
    c = get_curve(curve)
    out = np.zeros(8 * c.fq_limbs, dtype=np.uint64)
    inf = np.zeros(3, dtype=np.uint8)
    rm, sm = fr_to_mont([r], c)[0], fr_to_mont([s], c)[0]
    _lib.check(ctx.lib.zkp_groth16_assemble(ctx.h, c.cid, _ptr(np.ascontiguousarray(sums_xyz, dtype=np.uint64)),
                                            _ptr(rm), _ptr(sm), _ptr(out), _ptr(inf)), "zkp_groth16_assemble")
    return out, inf


def partials_bytes(ctx: Context, curve) -> int:
    n = C.c_size_t()
    _lib.check(ctx.lib.zkp_groth16_partials_bytes(get_curve(curve).cid, C.byref(n)), "zkp_groth16_partials_bytes")
    return n.value


def fold_assemble_dev(ctx: Context, curve, gathered_dev: int, world: int, r: int, s: int):
    """zkp_groth16_fold_assemble_dev: the all-gathered partial sums (device pointer, world x partials_bytes) ->
    (proof limbs, identity flags)."""
    c = get_curve(curve)
    out = np.zeros(8 * c.fq_limbs, dtype=np.uint64)
    inf = np.zeros(3, dtype=np.uint8)
    rm, sm = fr_to_mont([r], c)[0], fr_to_mont([s], c)[0]
    _lib.check(ctx.lib.zkp_groth16_fold_assemble_dev(ctx.h, c.cid, C.c_void_p(gathered_dev), world, _ptr(rm), _ptr(sm),
                                                     _ptr(out), _ptr(inf)), "zkp_groth16_fold_assemble_dev")
    return out, inf


class ProvingKey:
    """Device-resident proving key + circuit matrices (zkp_groth16_pk_upload).
    matrices_only=True uploads just at/bt/ct (witness_map only) for the host-driven sharded prover;
    shard=(rank, world) keeps 1/world of every query resident (zkp_groth16_pk_upload_shard): such a key yields the
    partial sums of its slices (partials_dev) and refuses to prove on its own."""

    def __init__(self, ctx: Context, params: Parameters, circuit, matrices_only: bool = False, shard=None, keep_form: bool = False):
        """keep_form: zkp_groth16_pk_upload_ex(ZKP_PK_KEEP_FORM) — no evaluation-form transforms at upload (one-shot callers)"""
        self.ctx, self.params, self.curve = ctx, params, params.curve
        self.shard = shard
        if shard is not None and keep_form:
            # zkp_groth16_pk_upload_shard has no flags argument: a sharded key always takes the context's key form
            # (zkp_ctx_config.h_evaluation_form / c_fold select it per context).  Silently ignoring the request was ADVICE r5.
            raise ValueError("keep_form applies to unsharded keys (zkp_groth16_pk_upload_ex); for a sharded key create the "
                             "context with dict(h_evaluation_form=False, c_fold=False)")
        inst = _as_instance(self.curve, circuit, assign=False)
        assert (inst.num_inputs, inst.num_aux, inst.num_constraints()) == \
            (params.num_inputs, params.num_aux, params.num_constraints)
        d = _lib.Groth16PkDesc()
        d.curve, d.num_inputs, d.num_aux, d.num_constraints = self.curve.cid, inst.num_inputs, inst.num_aux, \
            inst.num_constraints()
        keep = []

        def P(a):
            a = np.ascontiguousarray(a)
            keep.append(a)
            return a.ctypes.data

        for name, which in (("at", "a"), ("bt", "b"), ("ct", "c")):
            rp, col, cf = inst.csr(which)
            m = getattr(d, name)
            m.row_ptr, m.col, m.coeff = P(rp.astype(np.uint32)), P(col.astype(np.uint32)), P(cf.astype(np.uint64))
        for name in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2"):
            setattr(d, name, P(getattr(params, name).astype(np.uint64)))
        for name, fld in (() if matrices_only else (("a", "a_query"), ("b_g1", "b_g1_query"), ("b_g2", "b_g2_query"),
                                                    ("h", "h_query"), ("l", "l_query"))):
            xy, inf = getattr(params, fld)
            setattr(d, f"{name}_query", P(xy.astype(np.uint64)))
            setattr(d, f"{name}_inf", P(inf.astype(np.uint8)) if len(inf) else None)
            setattr(d, f"{name}_len", len(inf))
        h = C.c_void_p()
        if shard is None and keep_form:
            _lib.check(ctx.lib.zkp_groth16_pk_upload_ex(ctx.h, C.byref(d), 1, C.byref(h)), "zkp_groth16_pk_upload_ex")
        elif shard is None:
            _lib.check(ctx.lib.zkp_groth16_pk_upload(ctx.h, C.byref(d), C.byref(h)), "zkp_groth16_pk_upload")
        else:
            _lib.check(ctx.lib.zkp_groth16_pk_upload_shard(ctx.h, C.byref(d), shard[0], shard[1], C.byref(h)),
                       "zkp_groth16_pk_upload_shard")
        self.h = h
        n = C.c_uint64()
        _lib.check(ctx.lib.zkp_groth16_domain_size(self.h, C.byref(n)), "zkp_groth16_domain_size")
        self.domain_size = n.value
        self.nz = inst.num_inputs + inst.num_aux

    def table_plan(self) -> dict:
        """zkp_groth16_pk_info: how the window tables of this key are laid out (window groups when they did not fit)."""
        info = (C.c_uint64 * 8)()
        _lib.check(self.ctx.lib.zkp_groth16_pk_info(self.ctx.h, self.h, info), "zkp_groth16_pk_info")
        return {"window_group": int(info[0]), "table_bytes": int(info[1]), "window_bits": int(info[2]), "windows": int(info[3]),
                "table_copies": int(info[4]), "window_bits_b": int(info[5]), "b1_reuses_b2_sort": bool(info[6] & 1),
                "l_reuses_a_sort": bool(info[6] & 2), "shared_level1_pass": bool(info[6] & 4),
                "h_evaluation_form": bool(info[7] & 1), "c_folded_into_l": bool(info[7] & 2), "l_h_bucket_chained": bool(info[7] & 4)}

    def free(self):
        if self.h:
            _lib.check(self.ctx.lib.zkp_groth16_pk_free(self.ctx.h, self.h), "zkp_groth16_pk_free")
            self.h = None

    # R1CStoQAP::witness_map (r1cs_to_qap.rs:113-172)
    def witness_map(self, z_mont: np.ndarray) -> np.ndarray:
        z = np.ascontiguousarray(z_mont, dtype=np.uint64)
        assert z.shape == (self.nz, 4)
        h = np.zeros((self.domain_size, 4), dtype=np.uint64)
        _lib.check(self.ctx.lib.zkp_groth16_witness_map(self.ctx.h, self.h, _ptr(z), _ptr(h)), "zkp_groth16_witness_map")
        return h

    def witness_map_dev(self, z_dev: int, h_dev: int):
        """device pointers in and out (domain_size Fr written at h_dev)"""
        _lib.check(self.ctx.lib.zkp_groth16_witness_map_dev(self.ctx.h, self.h, C.c_void_p(z_dev), C.c_void_p(h_dev)),
                   "zkp_groth16_witness_map_dev")

    def partials_dev(self, z_dev: int, r: int, s: int, out_dev: int):
        """sharded key: witness map + the five partial MSMs of this rank -> partials_bytes() bytes at out_dev (device)"""
        c = self.curve
        rm, sm = fr_to_mont([r], c)[0], fr_to_mont([s], c)[0]
        _lib.check(self.ctx.lib.zkp_groth16_prove_partials_dev(self.ctx.h, self.h, C.c_void_p(z_dev), _ptr(rm), _ptr(sm),
                                                              C.c_void_p(out_dev)), "zkp_groth16_prove_partials_dev")

    def prove_raw(self, z, r_mont: np.ndarray, s_mont: np.ndarray, z_on_device: bool = False):
        """-> (proof limbs uint64, identity flags[3])."""
        c = self.curve
        words = 4 * c.fq_limbs + 4 * c.fq_limbs
        out = np.zeros(words, dtype=np.uint64)
        inf = np.zeros(3, dtype=np.uint8)
        r_mont = np.ascontiguousarray(r_mont, dtype=np.uint64)
        s_mont = np.ascontiguousarray(s_mont, dtype=np.uint64)
        if z_on_device:
            st = self.ctx.lib.zkp_groth16_prove_dev(self.ctx.h, self.h, C.c_void_p(z), _ptr(r_mont), _ptr(s_mont),
                                                    _ptr(out), _ptr(inf))
        else:
            z = np.ascontiguousarray(z, dtype=np.uint64)
            assert z.shape == (self.nz, 4)
            st = self.ctx.lib.zkp_groth16_prove(self.ctx.h, self.h, _ptr(z), _ptr(r_mont), _ptr(s_mont), _ptr(out),
                                                _ptr(inf))
        _lib.check(st, "zkp_groth16_prove")
        return out, inf

    def witness_map_host(self, z_mont: np.ndarray) -> np.ndarray:
        return self.witness_map(z_mont)

    def prove_batch_raw(self, z_devs, r_mont: np.ndarray, s_mont: np.ndarray, z_on_device: bool = True):
        """zkp_groth16_prove_batch_dev (z_devs: device pointers) / zkp_groth16_prove_batch (z_devs: HOST addresses of
        nz x 4 u64 buffers, ideally pinned): n proofs pipelined over the lanes -> (n x proof limbs, n x 3 flags)."""
        n = len(z_devs)
        words = 8 * self.curve.fq_limbs
        out = np.zeros((n, words), dtype=np.uint64)
        inf = np.zeros((n, 3), dtype=np.uint8)
        zp = (C.c_void_p * n)(*[C.c_void_p(z) for z in z_devs])
        r_mont = np.ascontiguousarray(r_mont, dtype=np.uint64).reshape(n, 4)
        s_mont = np.ascontiguousarray(s_mont, dtype=np.uint64).reshape(n, 4)
        fn = self.ctx.lib.zkp_groth16_prove_batch_dev if z_on_device else self.ctx.lib.zkp_groth16_prove_batch
        _lib.check(fn(self.ctx.h, self.h, n, C.cast(zp, C.c_void_p), _ptr(r_mont), _ptr(s_mont), _ptr(out), _ptr(inf)),
                   "zkp_groth16_prove_batch(_dev)")
        return out, inf

    def decode_proof(self, out: np.ndarray, inf) -> Proof:
        c = self.curve
        f = c.fq_limbs
        a = g1_from_mont(out[0:2 * f], [inf[0]], c)[0]
        b = g2_from_mont(out[2 * f:6 * f], [inf[1]], c)[0]
        cc = g1_from_mont(out[6 * f:8 * f], [inf[2]], c)[0]
        return Proof(a, b, cc)

    def last_timing(self) -> dict:
        t = _lib.Groth16Timing()
        _lib.check(self.ctx.lib.zkp_groth16_last_timing(self.ctx.h, C.byref(t)), "zkp_groth16_last_timing")
        return dict(ms_total=t.ms_total, ms_witness_map=t.ms_witness_map, ms_msm=list(t.ms_msm),
                    ms_assemble=t.ms_assemble, ms_msm_accumulate=t.ms_msm_accumulate,
                    msm_accumulate_launches=t.msm_accumulate_launches, msm_points=t.msm_points,
                    ms_msm_scan=t.ms_msm_scan, msm_scan_launches=t.msm_scan_launches, msm_scan_bytes=t.msm_scan_bytes,
                    ms_msm_acc=list(t.ms_msm_acc), msm_entries=list(t.msm_entries))


def _fill_desc(params: Parameters, inst: R1csInstance, matrices_only: bool = False):
    """zkp_groth16_pk_desc over host arrays (kept alive by the returned list)."""
    d = _lib.Groth16PkDesc()
    d.curve, d.num_inputs, d.num_aux, d.num_constraints = params.curve.cid, inst.num_inputs, inst.num_aux, \
        inst.num_constraints()
    keep = []

    def P(a):
        a = np.ascontiguousarray(a)
        keep.append(a)
        return a.ctypes.data

    for name, which in (("at", "a"), ("bt", "b"), ("ct", "c")):
        rp, col, cf = inst.csr(which)
        m = getattr(d, name)
        m.row_ptr, m.col, m.coeff = P(rp.astype(np.uint32)), P(col.astype(np.uint32)), P(cf.astype(np.uint64))
    for name in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2"):
        setattr(d, name, P(getattr(params, name).astype(np.uint64)))
    for name, fld in (() if matrices_only else (("a", "a_query"), ("b_g1", "b_g1_query"), ("b_g2", "b_g2_query"),
                                                ("h", "h_query"), ("l", "l_query"))):
        xy, inf = getattr(params, fld)
        setattr(d, f"{name}_query", P(xy.astype(np.uint64)))
        setattr(d, f"{name}_inf", P(inf.astype(np.uint8)) if len(inf) else None)
        setattr(d, f"{name}_len", len(inf))
    return d, keep


MULTI_SHARD, MULTI_REPLICATE = 0, 1


class MultiProvingKey:
    """zkp_groth16_pk_upload_multi on a MultiContext: mode MULTI_SHARD (every query split by index over the devices; one
    proof uses all of them: prove) or MULTI_REPLICATE (whole key per device; independent proofs round-robin: prove_batch)."""

    def __init__(self, mctx, params: Parameters, circuit, mode: int):
        self.ctx, self.params, self.curve, self.mode = mctx, params, params.curve, mode
        inst = _as_instance(self.curve, circuit, assign=False)
        d, keep = _fill_desc(params, inst)
        h = C.c_void_p()
        _lib.check(mctx.lib.zkp_groth16_pk_upload_multi(mctx.h, C.byref(d), mode, C.byref(h)), "zkp_groth16_pk_upload_multi")
        self.h = h
        self.nz = inst.num_inputs + inst.num_aux

    def free(self):
        if self.h:
            _lib.check(self.ctx.lib.zkp_groth16_pk_multi_free(self.ctx.h, self.h), "zkp_groth16_pk_multi_free")
            self.h = None

    def info(self) -> dict:
        """zkp_groth16_multi_info: what the last sharded proof did (exchange, witness-map variant and its measured times)"""
        v = (C.c_uint64 * 6)()
        _lib.check(self.ctx.lib.zkp_groth16_multi_info(self.ctx.h, self.h, v), "zkp_groth16_multi_info")
        return {"exchange": ("peer", "rccl", "peer (rccl watchdog gave up)")[min(int(v[0]), 2)], "rccl_ranks": int(v[1]),
                "witness_map": ("replicated", "split over devices 0..2", "measuring")[int(v[2])],
                "ms_replicated": v[3] / 1e3, "ms_split": v[4] / 1e3, "devices": int(v[5])}

    def prove_raw(self, z, r_mont, s_mont, z_on_device: bool = False):
        """SHARD key.  z: (nz, 4) host array, or — z_on_device — a list of one device pointer per rank."""
        c = self.curve
        out = np.zeros(8 * c.fq_limbs, dtype=np.uint64)
        inf = np.zeros(3, dtype=np.uint8)
        r_mont = np.ascontiguousarray(r_mont, dtype=np.uint64)
        s_mont = np.ascontiguousarray(s_mont, dtype=np.uint64)
        if z_on_device:
            zp = (C.c_void_p * len(z))(*[C.c_void_p(p) for p in z])
        else:
            z = np.ascontiguousarray(z, dtype=np.uint64)
            assert z.shape == (self.nz, 4)
            zp = (C.c_void_p * 1)(C.c_void_p(z.ctypes.data))
        _lib.check(self.ctx.lib.zkp_groth16_prove_multi(self.ctx.h, self.h, C.cast(zp, C.c_void_p), 1 if z_on_device else 0,
                                                        _ptr(r_mont), _ptr(s_mont), _ptr(out), _ptr(inf)),
                   "zkp_groth16_prove_multi")
        return out, inf

    def prove_batch_raw(self, zs, r_mont, s_mont, z_on_device: bool = False):
        """REPLICATE key.  zs[i]: host address (int) / host array of proof i's assignment, or — z_on_device — a device
        pointer on the device of rank i % n."""
        n = len(zs)
        out = np.zeros((n, 8 * self.curve.fq_limbs), dtype=np.uint64)
        inf = np.zeros((n, 3), dtype=np.uint8)
        keep = [np.ascontiguousarray(z, dtype=np.uint64) if not isinstance(z, int) else None for z in zs]
        ptrs = [z if isinstance(z, int) else k.ctypes.data for z, k in zip(zs, keep)]
        zp = (C.c_void_p * n)(*[C.c_void_p(p) for p in ptrs])
        r_mont = np.ascontiguousarray(r_mont, dtype=np.uint64).reshape(n, 4)
        s_mont = np.ascontiguousarray(s_mont, dtype=np.uint64).reshape(n, 4)
        _lib.check(self.ctx.lib.zkp_groth16_prove_batch_multi(self.ctx.h, self.h, n, C.cast(zp, C.c_void_p),
                                                              1 if z_on_device else 0, _ptr(r_mont), _ptr(s_mont),
                                                              _ptr(out), _ptr(inf)), "zkp_groth16_prove_batch_multi")
        return out, inf


def create_proof(pk: ProvingKey, circuit, r: int, s: int) -> Proof:
    """prover.rs:124-211: synthesise on the host, prove on the device."""
    inst = _as_instance(pk.curve, circuit, assign=True)
    z = fr_to_mont(inst.full_assignment(), pk.curve).reshape(-1, 4)
    rm = fr_to_mont([r], pk.curve)[0]
    sm = fr_to_mont([s], pk.curve)[0]
    out, inf = pk.prove_raw(z, rm, sm)
    return pk.decode_proof(out, inf)


def create_random_proof(pk: ProvingKey, circuit, rng=None) -> Proof:
    """prover.rs:97-111: r, s <- Fr::rand(rng)."""
    rnd = (lambda: rng.randrange(pk.curve.r)) if rng is not None else (lambda: secrets.randbelow(pk.curve.r))
    return create_proof(pk, circuit, rnd(), rnd())


def create_proof_no_zk(pk: ProvingKey, circuit) -> Proof:
    """prover.rs:113-122"""
    return create_proof(pk, circuit, 0, 0)
