"""ORACLE (test infrastructure only) — optimal-ate pairing on BN254 and BLS12-381, written from scratch, and
the Groth16 verifier of /root/reference/groth16/src/verifier.rs:18-44.

Purpose: reproduce the reference's OWN acceptance test (`verify_proof(..) == true`,
groth16/tests/mini.rs:89,96) on the proofs the oracle / the device produce, so the oracle is pinned by the same
predicate the reference's tests use (the reference holds no golden vectors).

Fp12 = Fp2[w]/(w^6 - xi) as a flat degree-6 extension of Fp2 (xi = 9+u for BN254, 1+u for BLS12-381);
slow but simple: schoolbook multiplication, final exponentiation by a plain square-and-multiply with the
exponent (q^12 - 1)/r (no Frobenius tricks, no inversion anywhere).  The Miller loop keeps T on the twist
E'(Fp2) in affine coordinates; vertical lines are dropped (denominator elimination, even embedding degree);
for the M-type twist the line is multiplied by w^3, an Fp4 element that the final exponentiation kills.
For BLS12-381 (negative x) the loop runs over |x| and the result is NOT conjugated: every pairing computed by
this module is therefore e(P,Q)^-1 on that curve — harmless for product-equals-one checks, which is the only
use here.
"""
from __future__ import annotations

from .curves import Group
from .fields import Curve, f2_add, f2_inv, f2_mul, f2_neg, f2_scalar, f2_sqr, f2_sub


class Fp12:
    def __init__(self, curve: Curve):
        self.p, self.xi = curve.q, curve.xi
        self.one = [(1, 0)] + [(0, 0)] * 5

    def mul(self, a, b):
        p = self.p
        t = [(0, 0)] * 11
        for i, ai in enumerate(a):
            if ai == (0, 0):
                continue
            for j, bj in enumerate(b):
                if bj == (0, 0):
                    continue
                t[i + j] = f2_add(t[i + j], f2_mul(ai, bj, p), p)
        out = list(t[:6])
        for k in range(6, 11):
            out[k - 6] = f2_add(out[k - 6], f2_mul(t[k], self.xi, p), p)
        return out

    def sqr(self, a):
        return self.mul(a, a)

    def pow(self, a, e: int):
        r = self.one
        for bit in bin(e)[2:]:
            r = self.sqr(r)
            if bit == "1":
                r = self.mul(r, a)
        return r


def _f2_pow(a, e, p):
    r = (1, 0)
    for bit in bin(e)[2:]:
        r = f2_sqr(r, p)
        if bit == "1":
            r = f2_mul(r, a, p)
    return r


class Pairing:
    def __init__(self, curve: Curve):
        self.c, self.p = curve, curve.q
        self.F12 = Fp12(curve)
        self.G1, self.G2 = Group(curve, 1), Group(curve, 2)
        self.final_exp = (curve.q ** 12 - 1) // curve.r
        if curve.name == "bn254":
            self.loop = 6 * curve.x_param + 2
            # Frobenius on the D-type twist: (x, y) -> (conj(x) * xi^((q-1)/3), conj(y) * xi^((q-1)/2))
            self.gx = _f2_pow(curve.xi, (curve.q - 1) // 3, curve.q)
            self.gy = _f2_pow(curve.xi, (curve.q - 1) // 2, curve.q)
        else:
            self.loop = curve.x_param

    # line through T (twist, affine) with slope lam, evaluated at P in G1 -> sparse Fp12
    def _line(self, T, lam, P):
        p = self.p
        xP, yP = P
        c = f2_sub(f2_mul(lam, T[0], p), T[1], p)          # lam*xT - yT
        d = f2_neg(f2_scalar(lam, xP, p), p)               # -lam*xP
        z = (0, 0)
        if self.c.twist_is_d:
            return [(yP, 0), d, z, c, z, z]                # yP - lam xP w + (lam xT - yT) w^3
        return [c, z, d, (yP, 0), z, z]                    # (lam xT - yT) - lam xP w^2 + yP w^3

    def _dbl(self, T, P):
        p = self.p
        lam = f2_mul(f2_scalar(f2_sqr(T[0], p), 3, p), f2_inv(f2_scalar(T[1], 2, p), p), p)
        l = self._line(T, lam, P)
        x3 = f2_sub(f2_sqr(lam, p), f2_scalar(T[0], 2, p), p)
        y3 = f2_sub(f2_mul(lam, f2_sub(T[0], x3, p), p), T[1], p)
        return (x3, y3), l

    def _add(self, T, Q, P):
        p = self.p
        lam = f2_mul(f2_sub(Q[1], T[1], p), f2_inv(f2_sub(Q[0], T[0], p), p), p)
        l = self._line(T, lam, P)
        x3 = f2_sub(f2_sub(f2_sqr(lam, p), T[0], p), Q[0], p)
        y3 = f2_sub(f2_mul(lam, f2_sub(T[0], x3, p), p), T[1], p)
        return (x3, y3), l

    def miller(self, P, Q):
        """P in G1 (affine ints), Q in G2 (affine Fp2 pairs); identity operand -> 1."""
        F = self.F12
        if P is None or Q is None:
            return F.one
        f, T = F.one, Q
        for bit in bin(self.loop)[3:]:
            T, l = self._dbl(T, P)
            f = F.mul(F.sqr(f), l)
            if bit == "1":
                T, l = self._add(T, Q, P)
                f = F.mul(f, l)
        if self.c.name == "bn254":
            p = self.p
            conj = lambda a: (a[0], (-a[1]) % p)
            Q1 = (f2_mul(conj(Q[0]), self.gx, p), f2_mul(conj(Q[1]), self.gy, p))
            Q2 = (f2_mul(conj(Q1[0]), self.gx, p), f2_mul(conj(Q1[1]), self.gy, p))
            Q2 = (Q2[0], f2_neg(Q2[1], p))
            T, l = self._add(T, Q1, P)
            f = F.mul(f, l)
            T, l = self._add(T, Q2, P)
            f = F.mul(f, l)
        return f

    def product_is_one(self, pairs) -> bool:
        """prod e(P_i, Q_i) == 1 ?"""
        F = self.F12
        f = F.one
        for P, Q in pairs:
            f = F.mul(f, self.miller(P, Q))
        return F.pow(f, self.final_exp) == F.one

    def pairing(self, P, Q):
        return self.F12.pow(self.miller(P, Q), self.final_exp)


def verify_proof(curve: Curve, vk, proof, public_inputs) -> bool:
    """groth16/src/verifier.rs:18-44: e(A,B) == e(alpha,beta) * e(sum_i x_i gamma_abc_i, gamma) * e(C, delta),
    checked as e(A,B) * e(g_ic, -gamma) * e(C, -delta) * e(-alpha, beta) == 1.
    vk: object with alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1 (oracle Parameters works)."""
    pr = Pairing(curve)
    G1, G2 = pr.G1, pr.G2
    if len(public_inputs) + 1 != len(vk.gamma_abc_g1):
        raise ValueError("MalformedVerifyingKey")
    g_ic = vk.gamma_abc_g1[0]
    for x, b in zip(public_inputs, vk.gamma_abc_g1[1:]):
        g_ic = G1.add(g_ic, G1.mul(b, x))
    return pr.product_is_one([(proof.a, proof.b), (g_ic, G2.neg(vk.gamma_g2)), (proof.c, G2.neg(vk.delta_g2)),
                              (G1.neg(vk.alpha_g1), vk.beta_g2)])
