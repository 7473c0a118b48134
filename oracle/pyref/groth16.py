"""ORACLE (test infrastructure only) — R1CS, QAP reduction and the Groth16 setup/prover, big-int Python.

Follows, function for function:
  /root/reference/r1cs/src/constraint_system.rs:10-93      ConstraintSystem / ConstraintSynthesizer
  /root/reference/groth16/src/generator.rs:38-132,135-286  KeypairAssembly, generate_parameters
  /root/reference/groth16/src/r1cs_to_qap.rs:16-52,58-110,113-172   evaluate_constraint, instance_map, witness_map
  /root/reference/groth16/src/prover.rs:16-95,124-211,213-228       ProvingAssignment, create_proof, calculate_coeff
The reference samples tau and the G1/G2 generators from an RNG (generator.rs:168,201-202); here they are
explicit inputs (there is no reference binary whose RNG stream could be matched).
"""
from __future__ import annotations

from dataclasses import dataclass, field

from .curves import Group
from .fields import Curve
from .ntt import Domain

INPUT, AUX = 0, 1


class LC:
    """LinearCombination: list of (Variable, coeff); Variable = (kind, index)."""

    def __init__(self, terms=None):
        self.terms = list(terms or [])

    def __add__(self, other):
        if isinstance(other, tuple) and len(other) == 2 and isinstance(other[1], tuple):
            coeff, var = other                      # lc + (coeff, var)
            return LC(self.terms + [(var, coeff)])
        return LC(self.terms + [(other, 1)])        # lc + var

    def __sub__(self, other):
        if isinstance(other, tuple) and len(other) == 2 and isinstance(other[1], tuple):
            coeff, var = other
            return LC(self.terms + [(var, -coeff)])
        return LC(self.terms + [(other, -1)])


class ConstraintSystem:
    ONE = (INPUT, 0)

    def __init__(self, curve: Curve, want_values: bool):
        self.curve, self.r, self.want_values = curve, curve.r, want_values
        self.at, self.bt, self.ct = [], [], []
        self.input_assignment, self.aux_assignment = [], []
        self.num_inputs = self.num_aux = 0
        self.alloc_input(lambda: 1)                 # the "one" input (prover.rs:143 / generator.rs:160)

    @staticmethod
    def one():
        return ConstraintSystem.ONE

    def alloc(self, f):
        if self.want_values:
            self.aux_assignment.append(f() % self.r)
        self.num_aux += 1
        return (AUX, self.num_aux - 1)

    def alloc_input(self, f):
        if self.want_values:
            self.input_assignment.append(f() % self.r)
        self.num_inputs += 1
        return (INPUT, self.num_inputs - 1)

    def enforce(self, a, b, c):
        for fn, rows in ((a, self.at), (b, self.bt), (c, self.ct)):
            lc = fn(LC())
            rows.append([(coeff % self.r, var) for var, coeff in lc.terms])   # push_constraints, lib.rs:128-139

    def num_constraints(self):
        return len(self.at)


def evaluate_constraint(terms, assignment, num_inputs, r):
    """r1cs_to_qap.rs:16-52."""
    acc = 0
    for coeff, (kind, i) in terms:
        acc += coeff * assignment[i if kind == INPUT else num_inputs + i]
    return acc % r


@dataclass
class Parameters:
    """groth16/src/lib.rs:79-91 (vk inlined: alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1)."""
    curve: Curve
    alpha_g1: tuple
    beta_g1: tuple
    beta_g2: tuple
    gamma_g2: tuple
    delta_g1: tuple
    delta_g2: tuple
    gamma_abc_g1: list
    a_query: list
    b_g1_query: list
    b_g2_query: list
    h_query: list
    l_query: list
    # oracle-only: exponents for the trapdoor check (never part of a real key)
    trapdoor: dict = field(default_factory=dict)


@dataclass
class Proof:
    a: tuple
    b: tuple
    c: tuple


def instance_map_with_evaluation(cs: ConstraintSystem, t: int):
    """r1cs_to_qap.rs:58-110."""
    r = cs.r
    domain = Domain(cs.curve, cs.num_constraints() + (cs.num_inputs - 1) + 1)
    zt = domain.evaluate_vanishing_polynomial(t)
    u = domain.evaluate_all_lagrange_coefficients(t)
    nvars = (cs.num_inputs - 1) + cs.num_aux
    a, b, c = [0] * (nvars + 1), [0] * (nvars + 1), [0] * (nvars + 1)
    for i in range(cs.num_inputs):
        a[i] = u[cs.num_constraints() + i]
    for i in range(cs.num_constraints()):
        for rows, out in ((cs.at, a), (cs.bt, b), (cs.ct, c)):
            for coeff, (kind, j) in rows[i]:
                idx = j if kind == INPUT else cs.num_inputs + j
                out[idx] = (out[idx] + u[i] * coeff) % r
    return a, b, c, zt, nvars, domain.size


def generate_parameters(curve: Curve, circuit, alpha, beta, gamma, delta, tau, g1_k=1, g2_k=1) -> Parameters:
    """generator.rs:135-286 with explicit toxic waste; generators = g1_k*G1, g2_k*G2 (standard gens)."""
    r = curve.r
    G1, G2 = Group(curve, 1), Group(curve, 2)
    cs = ConstraintSystem(curve, want_values=False)
    circuit.generate_constraints(cs)
    a, b, c, zt, nvars, m_raw = instance_map_with_evaluation(cs, tau)
    gi, di = pow(gamma, -1, r), pow(delta, -1, r)
    gamma_abc = [(beta * a[i] + alpha * b[i] + c[i]) * gi % r for i in range(cs.num_inputs)]
    l = [(beta * a[i] + alpha * b[i] + c[i]) * di % r for i in range(nvars + 1)]
    h = [zt * di % r * pow(tau, i, r) % r for i in range(m_raw - 1)]
    g1, g2 = G1.mul(G1.gen, g1_k), G2.mul(G2.gen, g2_k)
    m1 = lambda k: G1.mul(g1, k)
    m2 = lambda k: G2.mul(g2, k)
    return Parameters(
        curve=curve, alpha_g1=m1(alpha), beta_g1=m1(beta), beta_g2=m2(beta), gamma_g2=m2(gamma),
        delta_g1=m1(delta), delta_g2=m2(delta), gamma_abc_g1=[m1(k) for k in gamma_abc],
        a_query=[m1(k) for k in a], b_g1_query=[m1(k) for k in b], b_g2_query=[m2(k) for k in b],
        h_query=[m1(k) for k in h], l_query=[m1(k) for k in l[cs.num_inputs:]],
        trapdoor=dict(alpha=alpha, beta=beta, gamma=gamma, delta=delta, tau=tau, g1_k=g1_k, g2_k=g2_k,
                      a=a, b=b, c=c, l=l, h=h, zt=zt),
    )


def witness_map(cs: ConstraintSystem):
    """r1cs_to_qap.rs:113-172 → h coefficients (length = domain size)."""
    r = cs.r
    ni, nc = cs.num_inputs, cs.num_constraints()
    full = cs.input_assignment + cs.aux_assignment
    domain = Domain(cs.curve, nc + ni)
    n = domain.size
    a, b, c = [0] * n, [0] * n, [0] * n
    for i in range(nc):
        a[i] = evaluate_constraint(cs.at[i], full, ni, r)
        b[i] = evaluate_constraint(cs.bt[i], full, ni, r)
        c[i] = evaluate_constraint(cs.ct[i], full, ni, r)
    for i in range(ni):
        a[nc + i] = full[i]
    abc = dict(a=list(a), b=list(b), c=list(c))
    a = domain.coset_fft(domain.ifft(a))
    b = domain.coset_fft(domain.ifft(b))
    c = domain.coset_fft(domain.ifft(c))
    ab = [(x * y - z) % r for x, y, z in zip(a, b, c)]
    ab = domain.divide_by_vanishing_poly_on_coset(ab)
    return domain.coset_ifft(ab), abc


def create_proof(params: Parameters, circuit, r_: int, s_: int, msm="naive"):
    """prover.rs:124-211.  Returns (Proof, intermediates)."""
    curve = params.curve
    G1, G2 = Group(curve, 1), Group(curve, 2)
    cs = ConstraintSystem(curve, want_values=True)
    circuit.generate_constraints(cs)
    h, abc = witness_map(cs)
    assignment = cs.input_assignment[1:] + cs.aux_assignment
    do_msm = (lambda G, b, s: G.msm_pippenger(b, s)) if msm == "pippenger" else (lambda G, b, s: G.msm_naive(b, s))

    def calculate_coeff(G, initial, query, vk_param):      # prover.rs:213-228
        acc = do_msm(G, query[1:], assignment)
        res = G.add(initial, query[0])
        res = G.add(res, acc)
        return G.add(res, vk_param)

    g_a = calculate_coeff(G1, G1.mul(params.delta_g1, r_), params.a_query, params.alpha_g1)
    if r_ % curve.r != 0:
        g1_b = calculate_coeff(G1, G1.mul(params.delta_g1, s_), params.b_g1_query, params.beta_g1)
    else:
        g1_b = None
    g2_b = calculate_coeff(G2, G2.mul(params.delta_g2, s_), params.b_g2_query, params.beta_g2)
    h_acc = do_msm(G1, params.h_query, h)                   # min-len truncation: len(h_query) = N-1
    l_acc = do_msm(G1, params.l_query, cs.aux_assignment)
    g_c = G1.mul(g_a, s_)
    g_c = G1.add(g_c, G1.mul(g1_b, r_))
    g_c = G1.add(g_c, G1.neg(G1.mul(params.delta_g1, r_ * s_ % curve.r)))
    g_c = G1.add(g_c, l_acc)
    g_c = G1.add(g_c, h_acc)
    inter = dict(h=h, abc=abc, input_assignment=cs.input_assignment, aux_assignment=cs.aux_assignment, cs=cs)
    return Proof(a=g_a, b=g2_b, c=g_c), inter


def expected_proof_trapdoor(params: Parameters, cs: ConstraintSystem, h, r_: int, s_: int) -> Proof:
    """SURVEY.md §8(c).3 — compute the proof *in the exponent* from the toxic waste, then one scalar mul
    per element.  Independent of any MSM/NTT implementation except through h (checked separately)."""
    curve, t = params.curve, params.trapdoor
    r = curve.r
    G1, G2 = Group(curve, 1), Group(curve, 2)
    z = cs.input_assignment + cs.aux_assignment
    ni = cs.num_inputs
    A = (t["alpha"] + sum(zi * ai for zi, ai in zip(z, t["a"])) + r_ * t["delta"]) % r
    B = (t["beta"] + sum(zi * bi for zi, bi in zip(z, t["b"])) + s_ * t["delta"]) % r
    L = sum(zi * li for zi, li in zip(z[ni:], t["l"][ni:])) % r
    H = sum(hi * qi for hi, qi in zip(h, t["h"])) % r
    C = (s_ * A + r_ * B - r_ * s_ % r * t["delta"] + L + H) % r
    g1 = G1.mul(G1.gen, t["g1_k"])
    g2 = G2.mul(G2.gen, t["g2_k"])
    return Proof(a=G1.mul(g1, A), b=G2.mul(g2, B), c=G1.mul(g1, C))


# ------------------------------------------------------------------ circuits used by the reference's tests
class MiniCircuit:
    """groth16/tests/mini.rs:12-44: x*(y+2) = z, repeated `num` times; z is the public input."""

    def __init__(self, x=None, y=None, z=None, num=10):
        self.x, self.y, self.z, self.num = x, y, z, num

    def generate_constraints(self, cs):
        vx = cs.alloc(lambda: self.x)
        vy = cs.alloc(lambda: self.y)
        vz = cs.alloc_input(lambda: self.z)
        for _ in range(self.num):
            cs.enforce(lambda lc: lc + vx, lambda lc: lc + vy + (2, cs.one()), lambda lc: lc + vz)


class MimcChain:
    """marlin/examples/mimc.rs:15-119: `n` independent MiMC-5 permutations; per sample 12 aux, 10 constraints;
    the image is allocated with `alloc` (aux), so the only public input is the constant one."""
    ROUNDS = 5

    def __init__(self, curve: Curve, constants, preimages):
        self.r, self.constants, self.preimages = curve.r, list(constants), list(preimages)
        assert len(self.constants) == self.ROUNDS

    def generate_constraints(self, cs):
        r = self.r
        for (xl0, xr0) in self.preimages:
            xl_v, xr_v = xl0 % r, xr0 % r
            xl = cs.alloc(lambda: xl_v)
            xr = cs.alloc(lambda: xr_v)
            for i in range(self.ROUNDS):
                ci = self.constants[i]
                tmp_v = (xl_v + ci) ** 2 % r
                tmp = cs.alloc(lambda: tmp_v)
                cs.enforce(lambda lc: lc + xl + (ci, cs.one()), lambda lc: lc + xl + (ci, cs.one()),
                           lambda lc: lc + tmp)
                new_v = ((xl_v + ci) * tmp_v + xr_v) % r
                new_xl = cs.alloc(lambda: new_v)
                cs.enforce(lambda lc: lc + tmp, lambda lc: lc + xl + (ci, cs.one()),
                           lambda lc: lc + new_xl - xr)
                xr, xr_v = xl, xl_v
                xl, xl_v = new_xl, new_v
