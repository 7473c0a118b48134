"""Thin object layer over the C ABI (numpy in / numpy out).  One `Context` per GPU / process."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .params import CurveParams, get_curve

NTT_FFT, NTT_IFFT, NTT_COSET_FFT, NTT_COSET_IFFT = 0, 1, 2, 3


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return C.c_void_p(a.ctypes.data)


def _c64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


class Context:
    """zkp_ctx: device, stream, twiddle tables, scratch, resident bases."""

    def __init__(self, device: int = 0, config=None):
        """config: None (defaults = environment) or a dict / _lib.CtxConfig of zkp_ctx_config fields for THIS context
        (zkp_ctx_create_ex), e.g. Context(0, dict(lanes=2, h_evaluation_form=False))."""
        self.lib = _lib.load()
        h = C.c_void_p()
        cfg = _lib.make_config(config)
        if cfg is None:
            _lib.check(self.lib.zkp_ctx_create(C.byref(h), device), "zkp_ctx_create")
        else:
            _lib.check(self.lib.zkp_ctx_create_ex(C.byref(h), device, C.byref(cfg)), "zkp_ctx_create_ex")
        self.h = h
        self.device = device

    def config(self) -> dict:
        """the resolved zkp_ctx_config of this context (zkp_ctx_get_config)"""
        cfg = _lib.CtxConfig()
        _lib.check(self.lib.zkp_ctx_get_config(self.h, C.byref(cfg)), "zkp_ctx_get_config")
        return {f[0]: getattr(cfg, f[0]) for f in _lib.CtxConfig._fields_}

    def close(self):
        if getattr(self, "h", None):
            if not getattr(self, "_borrowed", False):
                self.lib.zkp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- plumbing
    def set_stream(self, hip_stream: int | None):
        _lib.check(self.lib.zkp_ctx_set_stream(self.h, C.c_void_p(hip_stream or 0)), "zkp_ctx_set_stream")

    def sync(self):
        _lib.check(self.lib.zkp_ctx_sync(self.h), "zkp_ctx_sync")

    def set_profiling(self, on: bool):
        _lib.check(self.lib.zkp_set_profiling(self.h, 1 if on else 0), "zkp_set_profiling")

    def dev_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        _lib.check(self.lib.zkp_dev_alloc(self.h, nbytes, C.byref(p)), "zkp_dev_alloc")
        return p.value

    def dev_free(self, dptr: int):
        _lib.check(self.lib.zkp_dev_free(self.h, C.c_void_p(dptr)), "zkp_dev_free")

    def h2d(self, dptr: int, a: np.ndarray):
        a = np.ascontiguousarray(a)
        _lib.check(self.lib.zkp_h2d(self.h, C.c_void_p(dptr), _ptr(a), a.nbytes), "zkp_h2d")

    def d2h(self, a: np.ndarray, dptr: int):
        assert a.flags["C_CONTIGUOUS"]
        _lib.check(self.lib.zkp_d2h(self.h, _ptr(a), C.c_void_p(dptr), a.nbytes), "zkp_d2h")

    def to_device(self, a: np.ndarray) -> int:
        a = np.ascontiguousarray(a)
        d = self.dev_alloc(a.nbytes)
        self.h2d(d, a)
        return d

    def timer_start(self):
        _lib.check(self.lib.zkp_timer_start(self.h), "zkp_timer_start")

    def timer_stop_ms(self) -> float:
        ms = C.c_float()
        _lib.check(self.lib.zkp_timer_stop_ms(self.h, C.byref(ms)), "zkp_timer_stop_ms")
        return ms.value

    def bench_mulmod(self, curve, field: int, unsaturated: bool) -> float:
        """zkp_bench_mulmod: sustained 1e9 Montgomery products / s of the library's multiplier (field 0 = Fr, 1 = Fq)"""
        g = C.c_double()
        _lib.check(self.lib.zkp_bench_mulmod(self.h, get_curve(curve).cid, field, 1 if unsaturated else 0, C.byref(g)),
                   "zkp_bench_mulmod")
        return g.value

    def bench_hbm_copy(self, nbytes: int = 2 << 30) -> float:
        """zkp_bench_hbm_copy: GB/s (read + written) of a streaming copy kernel on this device"""
        g = C.c_double()
        _lib.check(self.lib.zkp_bench_hbm_copy(self.h, nbytes, C.byref(g)), "zkp_bench_hbm_copy")
        return g.value

    # ---- NTT (ark-poly EvaluationDomain ops)
    def ntt(self, curve, data: np.ndarray, op: int) -> np.ndarray:
        """data: (2^k, 4) uint64 Montgomery Fr; returns the transformed copy."""
        c = get_curve(curve)
        a = _c64(data).copy()
        n = a.shape[0]
        assert n & (n - 1) == 0 and a.shape[1] == 4
        _lib.check(self.lib.zkp_ntt(self.h, c.cid, _ptr(a), n.bit_length() - 1, op), "zkp_ntt")
        return a

    def ntt_dev(self, curve, dptr: int, log_n: int, op: int):
        _lib.check(self.lib.zkp_ntt_dev(self.h, get_curve(curve).cid, C.c_void_p(dptr), log_n, op), "zkp_ntt_dev")

    # ---- bases / MSM (ark-ec VariableBaseMSM)
    def upload_bases(self, curve, group: int, xy: np.ndarray, inf: np.ndarray | None = None) -> "Bases":
        c = get_curve(curve)
        xy = _c64(xy)
        n = xy.shape[0] if xy.ndim == 2 else 0
        if inf is not None:
            inf = np.ascontiguousarray(inf, dtype=np.uint8)
        hnd = C.c_uint64()
        fn = self.lib.zkp_bases_upload_g1 if group == 1 else self.lib.zkp_bases_upload_g2
        _lib.check(fn(self.h, c.cid, _ptr(xy), _ptr(inf), n, C.byref(hnd)), "zkp_bases_upload")
        return Bases(self, c, group, hnd.value, n)

    def msm_var(self, curve, group: int, xy: np.ndarray, inf, scalars: np.ndarray, montgomery: bool = False) -> np.ndarray:
        """zkp_msm_g*_var: true variable-base MSM (fresh host bases, nothing resident) -> Jacobian limbs."""
        c = get_curve(curve)
        xy, s = _c64(xy), _c64(scalars)
        n = min(xy.shape[0] if xy.ndim == 2 else 0, s.shape[0] if s.ndim == 2 else 0)
        if inf is not None:
            inf = np.ascontiguousarray(inf, dtype=np.uint8)
        out = np.zeros(3 * c.fq_limbs * (1 if group == 1 else 2), dtype=np.uint64)
        fn = self.lib.zkp_msm_g1_var if group == 1 else self.lib.zkp_msm_g2_var
        _lib.check(fn(self.h, c.cid, _ptr(xy), _ptr(inf), _ptr(s), n, 1 if montgomery else 0, _ptr(out)), "zkp_msm_var")
        return out

    def fold(self, curve, group: int, xyz: np.ndarray) -> np.ndarray:
        c = get_curve(curve)
        xyz = _c64(xyz)
        w = 3 * c.fq_limbs * (1 if group == 1 else 2)
        out = np.zeros(w, dtype=np.uint64)
        fn = self.lib.zkp_g1_fold if group == 1 else self.lib.zkp_g2_fold
        _lib.check(fn(self.h, c.cid, _ptr(xyz), xyz.size // w, _ptr(out)), "zkp_fold")
        return out

    def into_affine(self, curve, group: int, xyz: np.ndarray):
        c = get_curve(curve)
        w = 2 * c.fq_limbs * (1 if group == 1 else 2)
        xy = np.zeros(w, dtype=np.uint64)
        inf = np.zeros(1, dtype=np.uint8)
        fn = self.lib.zkp_g1_into_affine if group == 1 else self.lib.zkp_g2_into_affine
        _lib.check(fn(self.h, c.cid, _ptr(_c64(xyz)), _ptr(xy), _ptr(inf)), "zkp_into_affine")
        return xy, bool(inf[0])

    def decompress_points(self, curve, group: int, data: bytes):
        """zkp_g1/g2_decompress: ark-serialize compressed points -> ((n, w) affine Montgomery, (n,) identity flags).  Raises
        ValueError(index) on a malformed point."""
        c = get_curve(curve)
        pb = 8 * c.fq_limbs * (1 if group == 1 else 2)
        assert len(data) % pb == 0
        n = len(data) // pb
        w = 2 * c.fq_limbs * (1 if group == 1 else 2)
        xy = np.zeros((n, w), dtype=np.uint64)
        inf = np.zeros(n, dtype=np.uint8)
        buf = np.frombuffer(data, dtype=np.uint8)
        bad = C.c_size_t(0)
        fn = self.lib.zkp_g1_decompress if group == 1 else self.lib.zkp_g2_decompress
        rc = fn(self.h, c.cid, _ptr(buf) if n else None, n, _ptr(xy), _ptr(inf), C.byref(bad))
        if rc == _lib.ZKP_ERR_INVALID_POINT:
            raise ValueError(f"malformed compressed point at index {bad.value}")
        _lib.check(rc, "zkp_decompress")
        return xy, inf

    def compress_points(self, curve, group: int, xy: np.ndarray, inf=None) -> bytes:
        """zkp_g1/g2_compress: affine Montgomery points -> ark-serialize compressed bytes"""
        c = get_curve(curve)
        xy = _c64(xy)
        n = xy.shape[0]
        pb = 8 * c.fq_limbs * (1 if group == 1 else 2)
        out = np.zeros(n * pb, dtype=np.uint8)
        infa = None if inf is None else np.ascontiguousarray(inf, dtype=np.uint8)
        fn = self.lib.zkp_g1_compress if group == 1 else self.lib.zkp_g2_compress
        _lib.check(fn(self.h, c.cid, _ptr(xy), None if infa is None else _ptr(infa), n, _ptr(out)), "zkp_compress")
        return out.tobytes()

    def subgroup_check(self, curve, group: int, xy: np.ndarray, inf=None) -> None:
        """zkp_g1/g2_subgroup_check: every point on the curve and in the prime-order subgroup ([r]P = O), else ValueError(index) —
        the checked half of ark's `deserialize`."""
        c = get_curve(curve)
        xy = _c64(xy)
        n = xy.shape[0]
        infa = None if inf is None else np.ascontiguousarray(inf, dtype=np.uint8)
        bad = C.c_size_t(0)
        fn = self.lib.zkp_g1_subgroup_check if group == 1 else self.lib.zkp_g2_subgroup_check
        rc = fn(self.h, c.cid, _ptr(xy) if n else None, None if infa is None else _ptr(infa), n, C.byref(bad))
        if rc == _lib.ZKP_ERR_INVALID_POINT:
            raise ValueError(f"point {bad.value} is not in the prime-order subgroup (or not on the curve)")
        _lib.check(rc, "zkp_subgroup_check")

    def fixed_base_mul(self, curve, group: int, base_xy: np.ndarray, scalars: np.ndarray):
        """k_i * P for canonical scalars (n,4) -> ((n, w) uint64 affine Montgomery, (n,) uint8 identity flags)."""
        c = get_curve(curve)
        scalars = _c64(scalars)
        n = scalars.shape[0]
        w = 2 * c.fq_limbs * (1 if group == 1 else 2)
        out = np.zeros((n, w), dtype=np.uint64)
        inf = np.zeros(n, dtype=np.uint8)
        fn = self.lib.zkp_fixed_base_mul_g1 if group == 1 else self.lib.zkp_fixed_base_mul_g2
        _lib.check(fn(self.h, c.cid, _ptr(_c64(base_xy)), _ptr(scalars), n, _ptr(out), _ptr(inf)), "zkp_fixed_base_mul")
        return out, inf


class Bases:
    """Device-resident query / SRS (zkp_bases_upload_*)."""

    def __init__(self, ctx: Context, curve: CurveParams, group: int, handle: int, n: int):
        self.ctx, self.curve, self.group, self.handle, self.n = ctx, curve, group, handle, n

    def free(self):
        if self.handle:
            _lib.check(self.ctx.lib.zkp_bases_free(self.ctx.h, self.handle), "zkp_bases_free")
            self.handle = 0

    def share_with(self, other_ctx: "Context") -> "Bases":
        """the same resident tables, addressable through another context (one context per prover thread)"""
        h = C.c_uint64(0)
        _lib.check(self.ctx.lib.zkp_bases_share(other_ctx.h, self.ctx.h, self.handle, C.byref(h)), "zkp_bases_share")
        return Bases(other_ctx, self.curve, self.group, h.value, self.n)

    def _out(self):
        return np.zeros(3 * self.curve.fq_limbs * (1 if self.group == 1 else 2), dtype=np.uint64)

    def msm(self, scalars: np.ndarray, offset: int = 0) -> np.ndarray:
        """VariableBaseMSM::multi_scalar_mul(bases[offset..], scalars) -> Jacobian (X,Y,Z) Montgomery limbs."""
        s = _c64(scalars)
        n = s.shape[0] if s.ndim == 2 else 0
        out = self._out()
        fn = self.ctx.lib.zkp_msm_g1 if self.group == 1 else self.ctx.lib.zkp_msm_g2
        _lib.check(fn(self.ctx.h, self.handle, offset, _ptr(s), n, _ptr(out)), "zkp_msm")
        return out

    def msm_dev(self, scalars_dev: int, n: int, offset: int = 0) -> np.ndarray:
        out = self._out()
        fn = self.ctx.lib.zkp_msm_g1_dev if self.group == 1 else self.ctx.lib.zkp_msm_g2_dev
        _lib.check(fn(self.ctx.h, self.handle, offset, C.c_void_p(scalars_dev), n, _ptr(out)), "zkp_msm_dev")
        return out

    def vartime_multiscalar_mul(self, fr_scalars_mont: np.ndarray) -> np.ndarray:
        """zkp_curve::Curve::vartime_multiscalar_mul (curve/src/lib.rs:38-45): Montgomery Fr scalars."""
        s = _c64(fr_scalars_mont)
        out = self._out()
        fn = self.ctx.lib.zkp_vartime_multiscalar_mul_g1 if self.group == 1 else \
            self.ctx.lib.zkp_vartime_multiscalar_mul_g2
        _lib.check(fn(self.ctx.h, self.handle, _ptr(s), s.shape[0] if s.ndim == 2 else 0, _ptr(out)),
                   "zkp_vartime_multiscalar_mul")
        return out

    def msm_affine(self, scalars: np.ndarray, offset: int = 0):
        return self.ctx.into_affine(self.curve, self.group, self.msm(scalars, offset))


# ---- Fr vector / polynomial primitives (device pointers), see include/zkp_accel.h
VEC_MUL, VEC_ADD, VEC_SUB, VEC_SCALE, VEC_AXPY = 0, 1, 2, 3, 4


def _ctx_method(fn):
    setattr(Context, fn.__name__, fn)
    return fn


@_ctx_method
def fr_vec_op(self, curve, op: int, a_dev: int, b_dev: int | None, out_dev: int, n: int, k_mont: np.ndarray | None = None):
    k = None if k_mont is None else _c64(k_mont)
    _lib.check(self.lib.zkp_fr_vec_op_dev(self.h, get_curve(curve).cid, op, C.c_void_p(a_dev),
                                          C.c_void_p(b_dev or 0), _ptr(k), C.c_void_p(out_dev), n), "zkp_fr_vec_op_dev")


@_ctx_method
def fr_batch_inverse(self, curve, v_dev: int, n: int):
    _lib.check(self.lib.zkp_fr_batch_inverse_dev(self.h, get_curve(curve).cid, C.c_void_p(v_dev), n),
               "zkp_fr_batch_inverse_dev")


@_ctx_method
def poly_evaluate(self, curve, p_dev: int, n: int, z_mont: np.ndarray) -> np.ndarray:
    out = np.zeros(4, dtype=np.uint64)
    _lib.check(self.lib.zkp_poly_evaluate_dev(self.h, get_curve(curve).cid, C.c_void_p(p_dev), n, _ptr(_c64(z_mont)),
                                              _ptr(out)), "zkp_poly_evaluate_dev")
    return out


@_ctx_method
def poly_div_linear(self, curve, p_dev: int, n: int, z_mont: np.ndarray, q_dev: int) -> np.ndarray:
    out = np.zeros(4, dtype=np.uint64)
    _lib.check(self.lib.zkp_poly_div_linear_dev(self.h, get_curve(curve).cid, C.c_void_p(p_dev), n, _ptr(_c64(z_mont)),
                                                C.c_void_p(q_dev), _ptr(out)), "zkp_poly_div_linear_dev")
    return out


def _bases_msm_mont_dev(self, scalars_dev: int, n: int, offset: int = 0) -> np.ndarray:
    """MSM of device-resident Montgomery Fr scalars against bases[offset..] (KZG10 commit/open; sharded queries)."""
    out = self._out()
    fn = self.ctx.lib.zkp_msm_g1_mont_dev if self.group == 1 else self.ctx.lib.zkp_msm_g2_mont_dev
    _lib.check(fn(self.ctx.h, self.handle, offset, C.c_void_p(scalars_dev), n, _ptr(out)), "zkp_msm_mont_dev")
    return out


Bases.msm_mont_dev = _bases_msm_mont_dev


def _bases_msm_mont_batch_dev(self, jobs) -> np.ndarray:
    """PC::commit over a list: jobs = [(scalars_dev, n, offset)] -> (len(jobs), 3 * fq_limbs) Jacobian results; the MSMs
    run three at a time on the context's MSM streams."""
    assert self.group == 1
    k = len(jobs)
    out = np.zeros((k, 3 * self.curve.fq_limbs), dtype=np.uint64)
    if k == 0:
        return out
    ptrs = (C.c_void_p * k)(*[j[0] for j in jobs])
    ns = (C.c_size_t * k)(*[j[1] for j in jobs])
    offs = (C.c_size_t * k)(*[j[2] for j in jobs])
    _lib.check(self.ctx.lib.zkp_msm_g1_mont_batch_dev(self.ctx.h, self.handle, k, offs, ptrs, ns, _ptr(out)),
               "zkp_msm_g1_mont_batch_dev")
    return out


Bases.msm_mont_batch_dev = _bases_msm_mont_batch_dev


VEC_ADDC = 5


@_ctx_method
def fr_spmv(self, curve, row_ptr_dev: int, col_dev: int, coeff_dev: int, nrows: int, x_dev: int, out_dev: int):
    """out = M x for a CSR matrix resident in HBM (z_a = A z, and the transposed product behind Marlin's t)."""
    _lib.check(self.lib.zkp_fr_spmv_dev(self.h, get_curve(curve).cid, C.c_void_p(row_ptr_dev), C.c_void_p(col_dev),
                                        C.c_void_p(coeff_dev), nrows, C.c_void_p(x_dev), C.c_void_p(out_dev)),
               "zkp_fr_spmv_dev")


@_ctx_method
def fr_gather(self, in_dev: int, idx_dev: int, n: int, out_dev: int):
    _lib.check(self.lib.zkp_fr_gather_dev(self.h, C.c_void_p(in_dev), C.c_void_p(idx_dev), n, C.c_void_p(out_dev)),
               "zkp_fr_gather_dev")


@_ctx_method
def poly_divide_by_vanishing(self, curve, p_dev: int, length: int, n: int, q_dev: int | None, rem_dev: int | None):
    _lib.check(self.lib.zkp_poly_divide_by_vanishing_dev(self.h, get_curve(curve).cid, C.c_void_p(p_dev), length, n,
                                                         C.c_void_p(q_dev or 0), C.c_void_p(rem_dev or 0)),
               "zkp_poly_divide_by_vanishing_dev")


@_ctx_method
def d2d(self, dst_dev: int, src_dev: int, nbytes: int):
    _lib.check(self.lib.zkp_d2d(self.h, C.c_void_p(dst_dev), C.c_void_p(src_dev), nbytes), "zkp_d2d")


@_ctx_method
def dev_zero(self, dst_dev: int, nbytes: int):
    _lib.check(self.lib.zkp_dev_zero(self.h, C.c_void_p(dst_dev), nbytes), "zkp_dev_zero")


@_ctx_method
def msm_mont_multi_dev(self, jobs) -> list:
    """jobs = [(Bases, scalars_dev, n, offset)] with possibly different (G1 / G2) base vectors -> list of Jacobian limb
    arrays; the MSMs run four at a time on the context's MSM streams (zkp_msm_mont_multi_dev)."""
    k = len(jobs)
    if k == 0:
        return []
    slot = max(3 * b.curve.fq_limbs * (1 if b.group == 1 else 2) for b, *_ in jobs)
    out = np.zeros((k, slot), dtype=np.uint64)
    handles = (C.c_uint64 * k)(*[b.handle for b, *_ in jobs])
    ptrs = (C.c_void_p * k)(*[j[1] for j in jobs])
    ns = (C.c_size_t * k)(*[j[2] for j in jobs])
    offs = (C.c_size_t * k)(*[j[3] for j in jobs])
    _lib.check(self.lib.zkp_msm_mont_multi_dev(self.h, k, handles, offs, ptrs, ns, _ptr(out), slot), "zkp_msm_mont_multi_dev")
    return [out[i, :3 * b.curve.fq_limbs * (1 if b.group == 1 else 2)].copy() for i, (b, *_) in enumerate(jobs)]


class MultiContext(Context):
    """zkp_ctx_create_multi: ONE process, one context per listed device (ids may repeat on a one-GPU box), the exchange
    step of the base-sharded prover owned by the library.  The object itself is rank 0's context (usable like any
    Context); member(k) borrows rank k's."""

    def __init__(self, device_ids, config=None):
        self.lib = _lib.load()
        ids = (C.c_int * len(device_ids))(*device_ids)
        h = C.c_void_p()
        cfg = _lib.make_config(config)
        _lib.check(self.lib.zkp_ctx_create_multi_ex(C.byref(h), ids, len(device_ids), C.byref(cfg) if cfg is not None else None),
                   "zkp_ctx_create_multi_ex")
        self.h = h
        self.device = device_ids[0]
        self.device_ids = list(device_ids)
        n = C.c_int32()
        _lib.check(self.lib.zkp_ctx_num_devices(self.h, C.byref(n)), "zkp_ctx_num_devices")
        assert n.value == len(device_ids)

    @property
    def num_devices(self) -> int:
        return len(self.device_ids)

    def member(self, rank: int) -> Context:
        m = C.c_void_p()
        _lib.check(self.lib.zkp_ctx_device(self.h, rank, C.byref(m)), "zkp_ctx_device")
        c = Context.__new__(Context)
        c.lib, c.h, c.device, c._borrowed = self.lib, m, self.device_ids[rank], True
        return c
