"""Host-side mirror of the reference's KZG10 polynomial commitment on the MI355X backend
(/root/reference/marlin/src/pc/kzg10.rs:100-156,185-226; used by `PC::commit / open`, pc/mod.rs:34-100).

`setup` is test/bench infrastructure with an explicit trapdoor (the reference samples beta, g, gamma_g, h from an RNG,
kzg10.rs:27-31); `commit` and `open` are the prover-side hot path: coefficient vectors live in HBM, the witness
polynomial p/(X - z) and the evaluations are computed on the device, and the MSMs run against the resident powers
with `offset = number of leading zero coefficients` exactly as `skip_leading_zeros_and_convert_to_bigints` does.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import codec
from .api import Context
from .params import get_curve


class KzgError(Exception):
    """marlin/src/pc/error.rs"""


@dataclass
class CommitterKey:
    curve: object
    powers_of_g: object          # api.Bases (device resident)
    powers_of_gamma_g: object
    supported_degree: int
    host_g: tuple = None         # (xy, inf) kept for tests
    host_gamma_g: tuple = None
    vk_g2: tuple = None          # (h, beta_h) canonical G2 points: the VerifierKey's G2 half (pc/data_structures.rs:100-109)


def _device_powers(ctx: Context, c, n: int, beta: int, k: int) -> np.ndarray:
    """canonical limbs of k * beta^i, i < n, generated in HBM by log2(n) doubling steps (v[len..2len) = beta^len * v[0..len))"""
    buf = ctx.dev_alloc(n * 32)
    try:
        ctx.h2d(buf, codec.fr_to_mont([1], c).reshape(1, 4))
        ln = 1
        while ln < n:
            m = min(ln, n - ln)
            ctx.fr_vec_op(c, 3, buf, None, buf + ln * 32, m, codec.fr_to_mont([pow(beta, ln, c.r)], c)[0])
            ln += m
        # scaling by k / R leaves the canonical (non-Montgomery) representation of k * beta^i in memory
        r_inv = pow(1 << (64 * c.fr_limbs), -1, c.r)
        ctx.fr_vec_op(c, 3, buf, None, buf, n, codec.fr_to_mont([k * r_inv % c.r], c)[0])
        out = np.zeros((n, c.fr_limbs), dtype=np.uint64)
        ctx.d2h(out, buf)
    finally:
        ctx.dev_free(buf)
    return out


def setup(ctx: Context, curve, max_degree: int, beta: int, g_k: int = 1, gamma_k: int = 7, h_k: int = 1) -> CommitterKey:
    """KZG10::setup + trim with explicit toxic waste: powers_of_g[i] = beta^i * (g_k G), powers_of_gamma_g likewise;
    h = h_k * G2, beta_h = beta * h (kzg10.rs:27-60)."""
    c = get_curve(curve)
    base, _ = codec.g1_to_mont([c.g1], c)
    g = ctx.fixed_base_mul(c, 1, base, _device_powers(ctx, c, max_degree + 1, beta, g_k))
    gg = ctx.fixed_base_mul(c, 1, base, _device_powers(ctx, c, max_degree + 1, beta, gamma_k))
    base2, _ = codec.g2_to_mont([c.g2], c)
    h_xy, h_inf = ctx.fixed_base_mul(c, 2, base2, codec.fr_canonical([h_k % c.r, h_k * beta % c.r], c))
    vk_g2 = tuple(codec.g2_from_mont(h_xy, h_inf, c))
    return CommitterKey(c, ctx.upload_bases(c, 1, *g), ctx.upload_bases(c, 1, *gg), max_degree, g, gg, vk_g2)


def _leading_zeros(coeffs_mont: np.ndarray) -> int:
    nz = np.flatnonzero(coeffs_mont.any(axis=1))
    return int(nz[0]) if len(nz) else len(coeffs_mont)


def _degree(coeffs_mont: np.ndarray) -> int:
    nz = np.flatnonzero(coeffs_mont.any(axis=1))
    return int(nz[-1]) if len(nz) else 0


def commit(ctx: Context, ck: CommitterKey, coeffs_mont: np.ndarray, blinding_mont: np.ndarray | None = None,
           power_offset: int = 0):
    """KZG10::commit (kzg10.rs:100-123) -> affine commitment (canonical ints or None).
    coeffs_mont: (deg+1, 4) Montgomery Fr; blinding_mont: the `Rand` blinding polynomial (hiding) or None.
    power_offset > 0 commits against `ck.shifted_powers(degree_bound)` = powers_of_g[supported_degree - bound ..]
    (pc/data_structures.rs:88-99) for degree-bounded polynomials."""
    c = ck.curve
    deg = _degree(coeffs_mont)
    if deg < 1:
        raise KzgError("DegreeIsZero")
    if deg > ck.supported_degree - power_offset:
        raise KzgError("DegreeOutOfBound")
    lz = _leading_zeros(coeffs_mont)
    d = ctx.to_device(np.ascontiguousarray(coeffs_mont[lz:]))
    try:
        comm = ck.powers_of_g.msm_mont_dev(d, len(coeffs_mont) - lz, offset=power_offset + lz)
    finally:
        ctx.dev_free(d)
    if blinding_mont is not None:
        if len(blinding_mont) - 1 > ck.supported_degree + 1:
            raise KzgError("HidingBoundTooLarge")
        db = ctx.to_device(np.ascontiguousarray(blinding_mont))
        try:
            rc = ck.powers_of_gamma_g.msm_mont_dev(db, len(blinding_mont))
        finally:
            ctx.dev_free(db)
        comm = ctx.fold(c, 1, np.concatenate([comm, rc]))
    xy, inf = ctx.into_affine(c, 1, comm)
    return codec.g1_from_mont(xy, [inf], c)[0]


def open(ctx: Context, ck: CommitterKey, coeffs_mont: np.ndarray, point: int, blinding_mont: np.ndarray | None = None):
    """KZG10::open (kzg10.rs:125-156) -> (w affine, rand_v or None)."""
    c = ck.curve
    n = len(coeffs_mont)
    if _degree(coeffs_mont) < 1:
        raise KzgError("DegreeIsZero")
    z = codec.fr_to_mont([point], c)[0]

    def witness_msm(bases, poly):
        m = len(poly)
        dp = ctx.to_device(np.ascontiguousarray(poly))
        dq = ctx.dev_alloc(max(m - 1, 1) * 32)
        try:
            ev = ctx.poly_div_linear(c, dp, m, z, dq)
            q = np.zeros((m - 1, 4), dtype=np.uint64)
            ctx.d2h(q, dq)
            lz = _leading_zeros(q) if bases is ck.powers_of_g else 0
            out = bases.msm_mont_dev(dq + lz * 32, m - 1 - lz, offset=lz)
        finally:
            ctx.dev_free(dp)
            ctx.dev_free(dq)
        return out, ev

    w, _ = witness_msm(ck.powers_of_g, coeffs_mont)
    rand_v = None
    if blinding_mont is not None and blinding_mont.any():
        wb, ev = witness_msm(ck.powers_of_gamma_g, blinding_mont)
        w = ctx.fold(c, 1, np.concatenate([w, wb]))
        rand_v = codec.fr_from_mont(ev.reshape(1, 4), c)[0]
    xy, inf = ctx.into_affine(c, 1, w)
    return codec.g1_from_mont(xy, [inf], c)[0], rand_v
