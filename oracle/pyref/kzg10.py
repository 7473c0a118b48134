"""ORACLE (test infrastructure only) — KZG10 commit / open / check in big-int Python, following
/root/reference/marlin/src/pc/kzg10.rs:27-98 (setup, trim), :100-123 (commit), :125-156 (open), :158-173 (check),
:211-226 (compute_witness_polynomial).  beta and the generators are explicit (the reference samples them)."""
from __future__ import annotations

from .curves import Group
from .fields import Curve
from .pairing import Pairing


def setup(curve: Curve, max_degree: int, beta: int, g_k: int = 1, gamma_k: int = 7, h_k: int = 1):
    G1, G2 = Group(curve, 1), Group(curve, 2)
    g, gamma_g, h = G1.mul(G1.gen, g_k), G1.mul(G1.gen, gamma_k), G2.mul(G2.gen, h_k)
    pw = [pow(beta, i, curve.r) for i in range(max_degree + 1)]
    return dict(curve=curve, powers_of_g=[G1.mul(g, p) for p in pw], powers_of_gamma_g=[G1.mul(gamma_g, p) for p in pw],
                g=g, gamma_g=gamma_g, h=h, beta_h=G2.mul(h, beta))


def commit(pp, coeffs, blinding=None):
    G1 = Group(pp["curve"], 1)
    lz = 0
    while lz < len(coeffs) and coeffs[lz] == 0:
        lz += 1
    comm = G1.msm_naive(pp["powers_of_g"][lz:], coeffs[lz:])
    if blinding is not None:
        comm = G1.add(comm, G1.msm_naive(pp["powers_of_gamma_g"], blinding))
    return comm


def divide_by_linear(coeffs, z, r):
    """p / (X - z), remainder discarded (synthetic division)."""
    q, acc = [0] * (len(coeffs) - 1), 0
    for i in range(len(coeffs) - 1, 0, -1):
        acc = (coeffs[i] + z * acc) % r
        q[i - 1] = acc
    return q


def evaluate(coeffs, z, r):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * z + c) % r
    return acc


def open_(pp, coeffs, z, blinding=None):
    curve = pp["curve"]
    G1 = Group(curve, 1)
    wq = divide_by_linear(coeffs, z, curve.r)
    lz = 0
    while lz < len(wq) and wq[lz] == 0:
        lz += 1
    w = G1.msm_naive(pp["powers_of_g"][lz:], wq[lz:])
    rand_v = None
    if blinding is not None and any(blinding):
        w = G1.add(w, G1.msm_naive(pp["powers_of_gamma_g"], divide_by_linear(blinding, z, curve.r)))
        rand_v = evaluate(blinding, z, curve.r)
    return w, rand_v


def check(pp, comm, z, value, w, rand_v=None) -> bool:
    """e(comm - value*g - rand_v*gamma_g, h) == e(w, beta_h - z*h)   (kzg10.rs:158-173)"""
    curve = pp["curve"]
    pr = Pairing(curve)
    G1, G2 = pr.G1, pr.G2
    u = G1.add(comm, G1.neg(G1.mul(pp["g"], value)))
    if rand_v is not None:
        u = G1.add(u, G1.neg(G1.mul(pp["gamma_g"], rand_v)))
    v = G2.add(pp["beta_h"], G2.neg(G2.mul(pp["h"], z)))
    return pr.product_is_one([(u, pp["h"]), (G1.neg(w), v)])
