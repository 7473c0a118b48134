"""CPU: the Marlin/KZG10 oracle passes the reference's own acceptance tests (marlin/tests/mini.rs:81,87 ->
verify_proof == true; marlin/src/pc/kzg10.rs:229-270 -> KZG10::check) with a from-scratch pairing."""
import random

from oracle.pyref import groth16 as og
from oracle.pyref import kzg10 as K
from oracle.pyref import marlin as M
from oracle.pyref.fields import BN254


def test_kzg10_commit_open_check():
    C = BN254
    rnd = random.Random(2)
    pp = K.setup(C, 19, 0xABCDEF123)
    for hiding in (False, True):
        p = [rnd.randrange(C.r) for _ in range(20)]
        p[0] = 0
        blind = [rnd.randrange(C.r) for _ in range(3)] if hiding else None
        z = rnd.randrange(C.r)
        comm = K.commit(pp, p, blind)
        w, rv = K.open_(pp, p, z, blind)
        assert K.check(pp, comm, z, K.evaluate(p, z, C.r), w, rv)
        assert not K.check(pp, comm, z, (K.evaluate(p, z, C.r) + 1) % C.r, w, rv)


def test_marlin_mini_proof_verifies():
    """x*(y+2) = z with x=2, y=3, z=10 (marlin/tests/mini.rs:43-88), fixed challenges."""
    C = BN254
    r = C.r
    rnd = random.Random(1)
    circ = og.MiniCircuit(2, 3, 10, 10)
    idx = M.index(C, circ)
    pp = K.setup(C, idx["max_degree"], 0x1234567)
    H = idx["dh"].size
    R = dict(w=[rnd.randrange(r)], z_a=[rnd.randrange(r)], z_b=[rnd.randrange(r)],
             mask=[rnd.randrange(r) for _ in range(3 * H)],
             blind={l: [rnd.randrange(r), rnd.randrange(r)] for l in ("w", "z_a", "z_b", "g_1")},
             blind_shifted={"g_1": [rnd.randrange(r), rnd.randrange(r)]})
    ch = dict(alpha=rnd.randrange(r), eta_a=rnd.randrange(r), eta_b=rnd.randrange(r), eta_c=rnd.randrange(r),
              beta=rnd.randrange(r), gamma=rnd.randrange(r), xi=rnd.randrange(1 << 128))
    proof = M.create_proof(idx, pp, circ, R, ch)
    ic = M.index_commitments(idx, pp)
    assert M.verify_proof(idx, pp, ic, proof, [10], ch)
    assert not M.verify_proof(idx, pp, ic, proof, [11], ch)            # wrong public input
    bad = dict(proof, evaluations=proof["evaluations"][:-1] + [(proof["evaluations"][-1] + 1) % r])
    assert not M.verify_proof(idx, pp, ic, bad, [10], ch)


def test_marlin_random_proof_round_trip_with_derived_challenges():
    """The reference's own round trip (marlin/tests/mini.rs:81-88): create_random_proof -> verify_proof == true, with
    every verifier message DERIVED from the Fiat–Shamir transcript (fs_rng.rs; lib.rs:105-158 / :190-215): the verifier
    re-derives them from (ivk, public input, commitments, evaluations) only.  Changing the public input, a commitment or
    an evaluation changes the challenges and the proof is rejected."""
    C = BN254
    r = C.r
    rnd = random.Random(11)
    circ = og.MiniCircuit(2, 3, 10, 10)
    idx = M.index(C, circ)
    pp = K.setup(C, idx["max_degree"], 0x7654321)
    H = idx["dh"].size
    R = dict(w=[rnd.randrange(r)], z_a=[rnd.randrange(r)], z_b=[rnd.randrange(r)],
             mask=[rnd.randrange(r) for _ in range(3 * H)],
             blind={l: [rnd.randrange(r), rnd.randrange(r)] for l in ("w", "z_a", "z_b", "g_1")},
             blind_shifted={"g_1": [rnd.randrange(r), rnd.randrange(r)]})
    ic = M.index_commitments(idx, pp)
    proof = M.create_random_proof(idx, pp, ic, circ, R)
    wire = dict(commitments=proof["commitments"], evaluations=proof["evaluations"], opening_proofs=proof["opening_proofs"])
    assert M.verify_random_proof(idx, pp, ic, wire, [10])
    assert not M.verify_random_proof(idx, pp, ic, wire, [11])
    ch = proof["challenges"]
    assert pow(ch["alpha"], H, r) != 1 and pow(ch["beta"], H, r) != 1 and ch["xi"] < (1 << 128)
    bad = dict(wire, evaluations=wire["evaluations"][:-1] + [(wire["evaluations"][-1] + 1) % r])
    assert not M.verify_random_proof(idx, pp, ic, bad, [10])
    # the same proof under FIXED challenges equal to the derived ones is bit-identical (the transcript only chooses them)
    again = M.create_proof(idx, pp, circ, R, ch)
    assert again["commitments"] == proof["commitments"] and again["opening_proofs"] == proof["opening_proofs"]
    # a different zk mask changes the first commitments, hence every challenge
    R2 = dict(R, w=[(R["w"][0] + 1) % r])
    assert M.create_random_proof(idx, pp, ic, circ, R2)["challenges"]["alpha"] != ch["alpha"]
