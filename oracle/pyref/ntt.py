"""ORACLE (test infrastructure only) — radix-2 evaluation domain over Fr.

Mathematical contract of ark-poly 0.2 `GeneralEvaluationDomain` / `Radix2EvaluationDomain`
(third-party, absent from /root/reference; call sites groth16/src/r1cs_to_qap.rs:63-70,123-126,
144-169 and every interpolate/fft in marlin/src/ahp/prover.rs).  Values are canonical ints mod r.
"""
from __future__ import annotations

from .fields import Curve


class Domain:
    def __init__(self, curve: Curve, num_coeffs: int):
        """ark `EvaluationDomain::new(n)`: size = next_pow2(n); None (here: ValueError, mapped by callers to
        SynthesisError::PolynomialDegreeTooLarge, r1cs_to_qap.rs:123-125) if log2 > TWO_ADICITY."""
        size = 1
        while size < num_coeffs:
            size <<= 1
        self.log_size = size.bit_length() - 1
        if self.log_size > curve.two_adicity:
            raise ValueError("PolynomialDegreeTooLarge")
        self.curve, self.r, self.size = curve, curve.r, size
        self.group_gen = pow(curve.root_of_unity, 1 << (curve.two_adicity - self.log_size), self.r)
        self.group_gen_inv = pow(self.group_gen, -1, self.r)
        self.size_inv = pow(size, -1, self.r)
        self.coset_gen = curve.fr_generator          # ark `multiplicative_generator()`: 5 / 7
        self.coset_gen_inv = pow(self.coset_gen, -1, self.r)

    # -- core transform: iterative radix-2, natural order in / natural order out
    def _transform(self, a, w):
        n, r = self.size, self.r
        a = list(a) + [0] * (n - len(a))
        assert len(a) == n
        # bit-reversal then DIT
        j = 0
        for i in range(1, n):
            bit = n >> 1
            while j & bit:
                j ^= bit
                bit >>= 1
            j |= bit
            if i < j:
                a[i], a[j] = a[j], a[i]
        length = 2
        while length <= n:
            wl = pow(w, n // length, r)
            half = length >> 1
            for s in range(0, n, length):
                t = 1
                for k in range(s, s + half):
                    u, v = a[k], a[k + half] * t % r
                    a[k] = (u + v) % r
                    a[k + half] = (u - v) % r
                    t = t * wl % r
            length <<= 1
        return a

    def fft(self, coeffs):
        return self._transform(coeffs, self.group_gen)

    def ifft(self, evals):
        return [x * self.size_inv % self.r for x in self._transform(evals, self.group_gen_inv)]

    def coset_fft(self, coeffs):
        """distribute_powers(g) then fft."""
        r, g = self.r, self.coset_gen
        out, p = [], 1
        for c in list(coeffs) + [0] * (self.size - len(coeffs)):
            out.append(c * p % r)
            p = p * g % r
        return self.fft(out)

    def coset_ifft(self, evals):
        """ifft then distribute_powers(g^-1)."""
        r, gi = self.r, self.coset_gen_inv
        out, p = [], 1
        for c in self.ifft(evals):
            out.append(c * p % r)
            p = p * gi % r
        return out

    def dft_naive(self, a, inverse=False):
        """O(n^2) definition: out[i] = sum_j a[j] w^(ij) — independent cross-check of _transform."""
        n, r = self.size, self.r
        a = list(a) + [0] * (n - len(a))
        w = self.group_gen_inv if inverse else self.group_gen
        out = []
        for i in range(n):
            wi = pow(w, i, r)
            acc, t = 0, 1
            for j in range(n):
                acc += a[j] * t
                t = t * wi % r
            out.append(acc % r * (self.size_inv if inverse else 1) % r)
        return out

    # -- helpers
    def evaluate_vanishing_polynomial(self, t):
        return (pow(t, self.size, self.r) - 1) % self.r

    def elements(self):
        out, p = [], 1
        for _ in range(self.size):
            out.append(p)
            p = p * self.group_gen % self.r
        return out

    def evaluate_all_lagrange_coefficients(self, t):
        """L_i(t) = Z(t) w^i / (N (t - w^i)); unit vector if t is in the domain."""
        r, n = self.r, self.size
        z = self.evaluate_vanishing_polynomial(t)
        els = self.elements()
        if z == 0:
            return [1 if e == t % r else 0 for e in els]
        zn = z * self.size_inv % r
        return [zn * e % r * pow((t - e) % r, -1, r) % r for e in els]

    def divide_by_vanishing_poly_on_coset(self, evals):
        """multiply every element by (g^N - 1)^-1 (r1cs_to_qap.rs:168)."""
        i = pow(self.evaluate_vanishing_polynomial(self.coset_gen), -1, self.r)
        return [e * i % self.r for e in evals]
