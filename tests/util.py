"""Shared helpers for the parity tests: oracle <-> ABI conversions."""
import random

import numpy as np

from ckb_zkp_amd import codec
from ckb_zkp_amd.params import get_curve
from oracle.pyref import fields as ofields
from oracle.pyref.curves import Group

OC = {"bn254": ofields.BN254, "bls12_381": ofields.BLS12_381}
# ZKP_TEST_FULL=0: the full-size tests (2^24 Groth16, |H| = 2^20 Marlin, BLS12-381 2^22, the 2^24 kernel-level cases) shrink to one
# smaller representative each — same code paths, a quarter of the points — for a suite that has to fit a time limit.  Default: full.
import os
TEST_FULL = os.environ.get("ZKP_TEST_FULL", "1") != "0"


def jac_to_affine(ctx, curve, group, xyz):
    """device Jacobian limbs -> oracle-style affine point via zkp_g*_into_affine."""
    c = get_curve(curve)
    xy, inf = ctx.into_affine(c, group, xyz)
    if group == 1:
        return codec.g1_from_mont(xy, [inf], c)[0]
    return codec.g2_from_mont(xy, [inf], c)[0]


def jac_limbs_to_affine_oracle(curve, group, xyz):
    """Independent normalisation on the host with the oracle (does not trust the device's inversion)."""
    c = get_curve(curve)
    G = Group(OC[c.name], group)
    f = c.fq_limbs
    Ri = pow(1 << (64 * f), -1, c.q)
    v = [x * Ri % c.q for x in codec.limbs_to_ints(np.asarray(xyz).reshape(-1, f))]
    if group == 1:
        return G.to_affine((v[0], v[1], v[2]))
    return G.to_affine(((v[0], v[1]), (v[2], v[3]), (v[4], v[5])))


def random_points(curve, group, n, seed):
    G = Group(OC[get_curve(curve).name], group)
    rnd = random.Random(seed)
    return [G.mul(G.gen, rnd.randrange(1, G.order)) for _ in range(n)]


def to_abi_points(curve, group, pts):
    c = get_curve(curve)
    return codec.g1_to_mont(pts, c) if group == 1 else codec.g2_to_mont(pts, c)


def spot_check_qap_exponents(params, inst, samples=48, seed=5):
    """The trapdoor exponents a_i(tau), b_i(tau), c_i(tau), l_i, h_i a synthetic key was built from (generate_parameters: for large
    instances with the library's own Fr kernels) recomputed for a sample of variables WITHOUT any product code: the Lagrange
    coefficients u_k = L_k(tau) by oracle/cpu (inverse transform of the powers of tau; cross-checked against the closed form
    Z(tau) w^k / (N (tau - w^k)) in Python integers at a few k), a_i = sum_k A_ki u_k (+ u_{nc+i} for inputs) over the CSR arrays by
    oracle/cpu's fr_dot — the dense column of the constant ONE included — then l = (beta a + alpha b + c) / delta and
    h_i = Z(tau) tau^i / delta in Python integers (groth16/src/r1cs_to_qap.rs:58-110, generator.rs:205-256).  Keeps the
    trapdoor-in-the-exponent assertions independent of the kernels that produced the key."""
    from oracle import cpu_oracle
    c = params.curve
    oc = OC[c.name]
    r, t = c.r, params.toxic
    nc, ni, nv = inst.num_constraints(), inst.num_inputs, inst.num_inputs + inst.num_aux
    N = 1 << max(nc + ni - 1, 0).bit_length()
    lg = N.bit_length() - 1
    w = pow(pow(c.fr_generator, (r - 1) >> c.two_adicity, r), 1 << (c.two_adicity - lg), r)
    tau = t["tau"] % r
    zt = (pow(tau, N, r) - 1) % r
    assert zt == t["zt"] % r
    Ri = pow(1 << (64 * c.fr_limbs), -1, r)
    to_int = lambda row: int.from_bytes(np.asarray(row).tobytes(), "little") * Ri % r
    u = cpu_oracle.lagrange_coeffs(oc, lg, tau)
    rnd = random.Random(seed)
    zn = zt * pow(N, -1, r) % r
    for k in {0, 1, N - 1} | {rnd.randrange(N) for _ in range(6)}:
        assert to_int(u[k]) == zn * pow(w, k, r) % r * pow((tau - pow(w, k, r)) % r, -1, r) % r, ("u", k)
    idx = sorted(i for i in ({0, 1, ni - 1, ni, nv - 1} | {rnd.randrange(nv) for _ in range(samples)}) if 0 <= i < nv)
    mont = {k: getattr(t[k], "mont", None) for k in ("a", "b", "c", "l", "h")}
    get = lambda k, i: to_int(mont[k][i]) if mont[k] is not None else t[k][i] % r
    vals = {}
    for which in "abc":
        row_ptr, col, coeff = inst.csr(which)
        col = np.asarray(col)
        coeff = np.asarray(coeff, dtype=np.uint64).reshape(-1, 4)
        rows_of = np.repeat(np.arange(nc, dtype=np.int64), np.diff(np.asarray(row_ptr, dtype=np.int64)))
        for i in idx:
            e = np.flatnonzero(col == i)
            acc = cpu_oracle.fr_dot(oc, coeff[e], u[rows_of[e]]) if len(e) else 0
            if which == "a" and i < ni:
                acc += to_int(u[nc + i])
            vals[(which, i)] = acc % r
            assert get(which, i) == vals[(which, i)], (which, i)
    di = pow(t["delta"], -1, r)
    for i in idx:
        assert get("l", i) == (t["beta"] * vals[("a", i)] + t["alpha"] * vals[("b", i)] + vals[("c", i)]) * di % r, ("l", i)
    for i in {0, 1, 2, N // 2, N - 2} | {rnd.randrange(N - 1) for _ in range(8)}:
        if 0 <= i < N - 1:
            assert get("h", i) == zt * di % r * pow(tau, i, r) % r, ("h", i)


def trapdoor_proof_exponents(params, inst, z_mont, h_mont, r_, s_):
    """(A, B, C) of create_proof (groth16/src/prover.rs:164-210) computed IN THE EXPONENT from the toxic waste: the three
    inner products over the whole assignment / quotient by oracle/cpu's fr_dot (Montgomery arrays in, canonical int out)."""
    from oracle import cpu_oracle
    c = params.curve
    oc = OC[c.name]
    t, r, ni = params.toxic, c.r, inst.num_inputs
    mont = lambda k: t[k].mont if hasattr(t[k], "mont") else codec.fr_to_mont(list(t[k]), c).reshape(-1, 4)
    z_mont = np.asarray(z_mont).reshape(-1, 4)
    A = (t["alpha"] + cpu_oracle.fr_dot(oc, z_mont, mont("a")) + r_ * t["delta"]) % r
    B = (t["beta"] + cpu_oracle.fr_dot(oc, z_mont, mont("b")) + s_ * t["delta"]) % r
    L = cpu_oracle.fr_dot(oc, z_mont[ni:], mont("l")[ni:])
    H = cpu_oracle.fr_dot(oc, np.asarray(h_mont).reshape(-1, 4), mont("h"))
    return A, B, (s_ * A + r_ * B - r_ * s_ % r * t["delta"] + L + H) % r
