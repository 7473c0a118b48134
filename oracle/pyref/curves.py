"""ORACLE (test infrastructure only) — short-Weierstrass (a = 0) group arithmetic and MSM.

Mathematical contract of ark-ec 0.2 `short_weierstrass_jacobian::{GroupAffine, GroupProjective}` and
`msm::VariableBaseMSM::multi_scalar_mul` (third-party, absent from /root/reference; call sites:
groth16/src/prover.rs:187,190,220; marlin/src/pc/kzg10.rs:109,118,137,146; curve/src/lib.rs:44).

Affine point = (x, y) or None (identity).  Jacobian = (X, Y, Z), identity <=> Z == 0.
Coordinates are plain canonical integers (Fq) or (c0, c1) tuples (Fq2); see fields.FieldOps.
"""
from __future__ import annotations

from .fields import Curve, FieldOps


class Group:
    def __init__(self, curve: Curve, g: int):
        self.curve, self.g = curve, g
        self.F = FieldOps(curve.q, 1 if g == 1 else 2)
        self.b = curve.g1_b if g == 1 else curve.g2_b
        self.gen = curve.g1_gen if g == 1 else curve.g2_gen
        self.order = curve.r

    # ---- affine
    def on_curve(self, P) -> bool:
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.sqr(y) == F.add(F.mul(F.sqr(x), x), self.b)

    def neg(self, P):
        return None if P is None else (P[0], self.F.neg(P[1]))

    def add(self, P, Q):
        """Affine chord-and-tangent (ground truth; slow)."""
        F = self.F
        if P is None:
            return Q
        if Q is None:
            return P
        if P[0] == Q[0]:
            if P[1] != Q[1] or F.is_zero(P[1]):
                return None
            lam = F.mul(F.small(F.sqr(P[0]), 3), F.inv(F.small(P[1], 2)))
        else:
            lam = F.mul(F.sub(Q[1], P[1]), F.inv(F.sub(Q[0], P[0])))
        x3 = F.sub(F.sub(F.sqr(lam), P[0]), Q[0])
        y3 = F.sub(F.mul(lam, F.sub(P[0], x3)), P[1])
        return (x3, y3)

    # ---- Jacobian (used for speed in mul / MSM; formulas dbl-2009-l, add-2007-bl, madd-2007-bl)
    def to_jac(self, P):
        return (self.F.one, self.F.one, self.F.zero) if P is None else (P[0], P[1], self.F.one)

    def to_affine(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z):
            return None
        zi = F.inv(Z)
        zi2 = F.sqr(zi)
        return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))

    def jdbl(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z):
            return J
        A = F.sqr(X)
        B = F.sqr(Y)
        C = F.sqr(B)
        D = F.small(F.sub(F.sub(F.sqr(F.add(X, B)), A), C), 2)
        E = F.small(A, 3)
        X3 = F.sub(F.sqr(E), F.small(D, 2))
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), F.small(C, 8))
        Z3 = F.small(F.mul(Y, Z), 2)
        return (X3, Y3, Z3)

    def jadd(self, P, Q):
        F = self.F
        if F.is_zero(P[2]):
            return Q
        if F.is_zero(Q[2]):
            return P
        Z1Z1 = F.sqr(P[2])
        Z2Z2 = F.sqr(Q[2])
        U1 = F.mul(P[0], Z2Z2)
        U2 = F.mul(Q[0], Z1Z1)
        S1 = F.mul(F.mul(P[1], Q[2]), Z2Z2)
        S2 = F.mul(F.mul(Q[1], P[2]), Z1Z1)
        if U1 == U2:
            if S1 == S2:
                return self.jdbl(P)
            return (F.one, F.one, F.zero)
        H = F.sub(U2, U1)
        R = F.sub(S2, S1)
        HH = F.sqr(H)
        HHH = F.mul(H, HH)
        V = F.mul(U1, HH)
        X3 = F.sub(F.sub(F.sqr(R), HHH), F.small(V, 2))
        Y3 = F.sub(F.mul(R, F.sub(V, X3)), F.mul(S1, HHH))
        Z3 = F.mul(F.mul(P[2], Q[2]), H)
        return (X3, Y3, Z3)

    def jadd_mixed(self, P, Qa):
        """ark `add_assign_mixed`: identity operands ignored; equal operands → doubling."""
        if Qa is None:
            return P
        return self.jadd(P, (Qa[0], Qa[1], self.F.one))

    def jmul(self, J, k: int):
        k %= self.order
        R = (self.F.one, self.F.one, self.F.zero)
        for bit in bin(k)[2:] if k else "":
            R = self.jdbl(R)
            if bit == "1":
                R = self.jadd(R, J)
        return R

    def mul(self, P, k: int):
        return self.to_affine(self.jmul(self.to_jac(P), k))

    # ---- MSM
    def msm_naive(self, bases, scalars):
        """sum_i scalars[i]*bases[i] over min(len) pairs (ark semantics), double-and-add per term."""
        acc = (self.F.one, self.F.one, self.F.zero)
        for P, k in zip(bases, scalars):
            if P is None or k == 0:
                continue
            acc = self.jadd(acc, self.jmul(self.to_jac(P), k))
        return self.to_affine(acc)

    def msm_pippenger(self, bases, scalars):
        """Restatement of ark-ec 0.2 VariableBaseMSM: window c = 3 if n < 32 else ln_without_floats(n)+2
        where ln_without_floats(n) = log2(n)*69/100 (integer ops); zero scalars skipped, scalar == 1
        fast path in window 0, 2^c - 1 Jacobian buckets per window, running-sum reduction, Horner
        combine with c doublings from the top window down."""
        n = min(len(bases), len(scalars))
        bases, scalars = bases[:n], scalars[:n]
        c = ark_window_bits(n)
        num_bits = self.order.bit_length()
        zero = (self.F.one, self.F.one, self.F.zero)
        window_sums = []
        for w_start in range(0, num_bits, c):
            res = zero
            buckets = [zero] * ((1 << c) - 1)
            for P, k in zip(bases, scalars):
                if k == 0:
                    continue
                if k == 1:
                    if w_start == 0:
                        res = self.jadd_mixed(res, P)
                    continue
                d = (k >> w_start) % (1 << c)
                if d != 0:
                    buckets[d - 1] = self.jadd_mixed(buckets[d - 1], P)
            running = zero
            for b in reversed(buckets):
                running = self.jadd(running, b)
                res = self.jadd(res, running)
            window_sums.append(res)
        total = zero
        for ws in reversed(window_sums[1:]):
            total = self.jadd(total, ws)
            for _ in range(c):
                total = self.jdbl(total)
        total = self.jadd(total, window_sums[0])
        return self.to_affine(total)


def ark_window_bits(n: int) -> int:
    """c used by ark-ec 0.2 for an MSM of n terms."""
    if n < 32:
        return 3
    log2 = (n - 1).bit_length()          # ark_std::log2 = ceil(log2 n)
    return log2 * 69 // 100 + 2
