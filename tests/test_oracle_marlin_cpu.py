"""CPU: the C++ Marlin oracle (oracle/cpu/marlin_oracle.inc — the full-size checker and the timed cpu_baseline of
BASELINE.json configs[3]) against the big-int Python oracle (oracle/pyref/marlin.py) on small circuits: index polynomials and
commitments, every prover polynomial, commitment, evaluation and opening proof — with supplied verifier messages and with the
messages derived from the reference's Fiat–Shamir transcript — and the reference's own acceptance test (marlin/tests/mini.rs:81-87:
verify_proof == true, with a from-scratch pairing) on the C++ prover's output."""
import random

import pytest

from oracle import cpu_oracle
from oracle.pyref import groth16 as og
from oracle.pyref import kzg10 as K
from oracle.pyref import marlin as M
from oracle.pyref.fields import BLS12_381, BN254


def _rand(curve, hs, seed):
    r = curve.r
    rnd = random.Random(seed)
    R = dict(w=[rnd.randrange(r)], z_a=[rnd.randrange(r)], z_b=[rnd.randrange(r)], mask=[rnd.randrange(r) for _ in range(3 * hs)],
             blind={l: [rnd.randrange(r), rnd.randrange(r)] for l in ("w", "z_a", "z_b", "g_1")},
             blind_shifted={"g_1": [rnd.randrange(r), rnd.randrange(r)]})
    ch = dict(alpha=rnd.randrange(r), eta_a=rnd.randrange(r), eta_b=rnd.randrange(r), eta_c=rnd.randrange(r),
              beta=rnd.randrange(r), gamma=rnd.randrange(r), xi=rnd.randrange(1 << 128))
    return R, ch


class _SwapAB:
    """A and B exchanged: A becomes the denser matrix and balance_matrices swaps rows"""

    def __init__(self, inner):
        self.inner = inner

    def generate_constraints(self, cs):
        class Proxy:
            def __getattr__(self, name):
                return getattr(cs, name)

            def enforce(self, a, b, c):
                return cs.enforce(b, a, c)
        self.inner.generate_constraints(Proxy())


def _circuit(curve, kind, rnd):
    if kind == "mini":                                       # marlin/tests/mini.rs: x * (y + 2) = z, z public
        return og.MiniCircuit(2, 3, 10, 10), [10]
    consts = [rnd.randrange(curve.r) for _ in range(5)]
    pre = [(rnd.randrange(curve.r), rnd.randrange(curve.r)) for _ in range(2)]
    circ = og.MimcChain(curve, consts, pre)
    return (_SwapAB(circ) if kind == "mimc_swapped" else circ), []


@pytest.mark.parametrize("curve,kind,threads", [(BN254, "mini", 1), (BN254, "mimc", 3), (BLS12_381, "mimc", 2), (BN254, "mimc_swapped", 8)])
def test_cpp_marlin_oracle_matches_python_oracle(curve, kind, threads):
    rnd = random.Random(7)
    circ, public = _circuit(curve, kind, rnd)
    idx = M.index(curve, circ)
    cs = og.ConstraintSystem(curve, want_values=True)
    circ.generate_constraints(cs)                            # as synthesised: no squaring / balancing / sorting
    inst = cpu_oracle.SynthesisedInstance(cs)
    pp = K.setup(curve, idx["max_degree"], 0x1D2C3B4A59687)
    co = cpu_oracle.MarlinOracle(curve, inst, srs=cpu_oracle.srs_from_pyref(pp), threads=threads)
    try:
        assert (co.xs, co.hs, co.ks, co.bs, co.max_degree, co.num_non_zeros) == \
            (idx["dx"].size, idx["dh"].size, idx["dk"].size, idx["db"].size, idx["max_degree"], idx["num_non_zeros"])
        for m in "abc":
            for k in ("row", "col", "val", "row_col"):
                assert co.poly(f"{m}_{k}") == M.trim(idx["star"][m]["polys"][k]), (m, k)
        ic = M.index_commitments(idx, pp)
        assert co.index_commitments() == ic
        R, ch = _rand(curve, idx["hs"] if "hs" in idx else idx["dh"].size, seed=19)
        # supplied verifier messages
        want = M.create_proof(idx, pp, circ, R, ch)
        got = co.create_proof(inst.x, inst.w, R, M.FixedChallenger(ch))
        for l in cpu_oracle.MARLIN_LABELS:
            assert co.poly(l) == want["polys"][l], l
            assert got["commitments"][l] == want["commitments"][l], l
        assert got["query"] == want["query"] and got["evaluations"] == want["evaluations"]
        assert got["opening_proofs"] == want["opening_proofs"]
        assert M.verify_proof(idx, pp, ic, got, public, ch)
        # marlin::create_random_proof: messages derived from the transcript (oracle/pyref/fs_rng.py on both sides)
        ivk = M.index_verifier_key(idx, pp, ic)
        want = M.create_random_proof(idx, pp, ic, circ, R)
        got = co.create_proof(inst.x, inst.w, R, M.FiatShamirChallenger(idx, ivk, inst.x[1:]))
        assert got["challenges"] == want["challenges"]
        assert got["commitments"] == want["commitments"] and got["evaluations"] == want["evaluations"]
        assert got["opening_proofs"] == want["opening_proofs"]
        wire = dict(commitments=got["commitments"], evaluations=got["evaluations"], opening_proofs=got["opening_proofs"])
        assert M.verify_random_proof(idx, pp, ic, wire, public)
        assert not M.verify_random_proof(idx, pp, ic, wire, [(p + 1) % curve.r for p in public] or [5])
        # sequential commitments (the reference's order of work) give the same proof
        co.set_options(threads=1, concurrent_commits=False)
        again = co.create_proof(inst.x, inst.w, R, M.FiatShamirChallenger(idx, ivk, inst.x[1:]))
        assert again["commitments"] == got["commitments"] and again["opening_proofs"] == got["opening_proofs"]
        assert sum(again["phase_seconds"].values()) > 0
    finally:
        co.free()
