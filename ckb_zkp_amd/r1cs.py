"""Host-side mirror of the reference's constraint-system interface for the prover path.

Mirrors /root/reference/r1cs/src/constraint_system.rs:10-93 (`ConstraintSystem::{alloc, alloc_input, enforce,
one}`, `ConstraintSynthesizer::generate_constraints`) and the two implementations the Groth16 path uses:
`ProvingAssignment` (/root/reference/groth16/src/prover.rs:16-95) and `KeypairAssembly`
(/root/reference/groth16/src/generator.rs:38-132).  Synthesis stays on the host (it is closure-driven in the
reference too); what crosses the C ABI is the CSR form of at/bt/ct plus the assignment.
"""
from __future__ import annotations

import numpy as np

from .codec import fr_to_mont
from .params import CurveParams, get_curve

INPUT, AUX = 0, 1


class SynthesisError(Exception):
    """r1cs/src/error.rs:7-24"""


class AssignmentMissing(SynthesisError):
    pass


class PolynomialDegreeTooLarge(SynthesisError):
    pass


class LinearCombination:
    def __init__(self, terms=None):
        self.terms = list(terms or [])          # (variable, coeff)

    def __add__(self, other):
        if isinstance(other, tuple) and len(other) == 2 and isinstance(other[1], tuple):
            coeff, var = other                   # lc + (coeff, var)
            return LinearCombination(self.terms + [(var, coeff)])
        return LinearCombination(self.terms + [(other, 1)])

    def __sub__(self, other):
        if isinstance(other, tuple) and len(other) == 2 and isinstance(other[1], tuple):
            coeff, var = other
            return LinearCombination(self.terms + [(var, -coeff)])
        return LinearCombination(self.terms + [(other, -1)])


class ConstraintSystem:
    """Both `ProvingAssignment` (assign=True) and `KeypairAssembly` (assign=False)."""

    def __init__(self, curve, assign: bool):
        self.curve: CurveParams = get_curve(curve)
        self.assign = assign
        self.at, self.bt, self.ct = [], [], []
        self.input_assignment, self.aux_assignment = [], []
        self.num_inputs = self.num_aux = 0
        self.alloc_input(lambda: 1)              # prover.rs:143 / generator.rs:160

    @staticmethod
    def one():
        return (INPUT, 0)

    def _value(self, f):
        v = f()
        if v is None:
            raise AssignmentMissing()
        return int(v) % self.curve.r

    def alloc(self, f):
        if self.assign:
            self.aux_assignment.append(self._value(f))
        self.num_aux += 1
        return (AUX, self.num_aux - 1)

    def alloc_input(self, f):
        if self.assign:
            self.input_assignment.append(self._value(f))
        self.num_inputs += 1
        return (INPUT, self.num_inputs - 1)

    def enforce(self, a, b, c):
        r = self.curve.r
        for fn, rows in ((a, self.at), (b, self.bt), (c, self.ct)):
            lc = fn(LinearCombination())
            rows.append([(coeff % r, var) for var, coeff in lc.terms])

    def num_constraints(self) -> int:
        return len(self.at)

    # ---- what crosses the ABI
    def full_assignment(self) -> list:
        return self.input_assignment + self.aux_assignment

    def csr(self, which: str):
        """(row_ptr u32, col u32, coeff (nnz,4) u64 Montgomery); col indexes input ++ aux."""
        rows = {"a": self.at, "b": self.bt, "c": self.ct}[which]
        row_ptr = np.zeros(len(rows) + 1, dtype=np.uint32)
        cols, coeffs = [], []
        for i, row in enumerate(rows):
            for coeff, (kind, j) in row:
                cols.append(j if kind == INPUT else self.num_inputs + j)
                coeffs.append(coeff)
            row_ptr[i + 1] = len(cols)
        return row_ptr, np.asarray(cols, dtype=np.uint32), fr_to_mont(coeffs, self.curve).reshape(-1, 4)


class R1csInstance:
    """Array form of a synthesised system (what `ProvingAssignment` holds after synthesis), used for large
    synthetic instances that are generated without per-constraint closures."""

    def __init__(self, curve, num_inputs, num_aux, num_constraints, csr_a, csr_b, csr_c, z=None):
        self.curve = get_curve(curve)
        self.num_inputs, self.num_aux, self.num_constraints_ = num_inputs, num_aux, num_constraints
        self._csr = {"a": csr_a, "b": csr_b, "c": csr_c}
        self.z = z                               # list[int] canonical, len num_inputs + num_aux (or None)

    @classmethod
    def from_cs(cls, cs: ConstraintSystem):
        return cls(cs.curve, cs.num_inputs, cs.num_aux, cs.num_constraints(), cs.csr("a"), cs.csr("b"), cs.csr("c"),
                   cs.full_assignment() if cs.assign else None)

    def num_constraints(self):
        return self.num_constraints_

    def csr(self, which):
        return self._csr[which]

    def full_assignment(self):
        return self.z
