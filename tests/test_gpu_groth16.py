"""Groth16 prover parity (config 1 of BASELINE.json and the reference's own Mini round trip,
/root/reference/groth16/tests/mini.rs:46-97): device proof == oracle create_proof == trapdoor-expected proof."""
import random

import pytest

from ckb_zkp_amd import codec, groth16
from ckb_zkp_amd.circuits import Mini, MimcChain, mimc_chain_instance
from oracle.pyref import groth16 as og
from tests.util import OC

pytestmark = pytest.mark.gpu
TOXIC = dict(alpha=0x1234567890ABCDEF1, beta=0xFEDCBA09876543211, gamma=0x1111111111111111111,
             delta=0x2222222222222222223, tau=0x3333333333333333335)


def _oracle_params(curve, circuit, g1_k=1, g2_k=1):
    return og.generate_parameters(OC[curve], circuit, TOXIC["alpha"], TOXIC["beta"], TOXIC["gamma"], TOXIC["delta"],
                                  TOXIC["tau"], g1_k, g2_k)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_mini_roundtrip(ctx, curve):
    """x=2, y=3, z=10, num=10 (mini.rs:53-76).  Key from the device's fixed-base kernels == oracle key;
    proof == oracle prover == proof computed in the exponent."""
    params = groth16.generate_parameters(ctx, curve, Mini(num=10), **TOXIC, g1_k=3, g2_k=5)
    opk = _oracle_params(curve, og.MiniCircuit(num=10), 3, 5)
    c = params.curve
    assert codec.g1_from_mont(*params.a_query, c) == opk.a_query
    assert codec.g1_from_mont(*params.b_g1_query, c) == opk.b_g1_query
    assert codec.g2_from_mont(*params.b_g2_query, c) == opk.b_g2_query
    assert codec.g1_from_mont(*params.h_query, c) == opk.h_query
    assert codec.g1_from_mont(*params.l_query, c) == opk.l_query
    pk = groth16.ProvingKey(ctx, params, Mini(num=10))
    try:
        for r_, s_ in ((77, 88), (0, 0), (0, 5), (c.r - 1, c.r - 2)):
            proof = groth16.create_proof(pk, Mini(2, 3, 10, 10), r_, s_)
            oproof, inter = og.create_proof(opk, og.MiniCircuit(2, 3, 10, 10), r_, s_)
            assert (proof.a, proof.b, proof.c) == (oproof.a, oproof.b, oproof.c), (r_, s_)
            exp = og.expected_proof_trapdoor(opk, inter["cs"], inter["h"], r_, s_)
            assert (proof.a, proof.b, proof.c) == (exp.a, exp.b, exp.c)
    finally:
        pk.free()


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_witness_map_matches_oracle(ctx, curve):
    """R1CStoQAP::witness_map (r1cs_to_qap.rs:113-172) on a 30-constraint MiMC chain and on Mini."""
    rnd = random.Random(4)
    inst = mimc_chain_instance(curve, 3, seed=99)
    circ = MimcChain(curve, inst.constants, inst.preimages)
    ocirc = og.MimcChain(OC[curve], inst.constants, inst.preimages)
    params = groth16.generate_parameters(ctx, curve, circ, **TOXIC)
    pk = groth16.ProvingKey(ctx, params, inst)
    try:
        cs = og.ConstraintSystem(OC[curve], True)
        ocirc.generate_constraints(cs)
        h_exp, _ = og.witness_map(cs)
        z = codec.fr_to_mont(inst.z, params.curve).reshape(-1, 4)
        assert codec.fr_from_mont(pk.witness_map(z), params.curve) == h_exp
        assert h_exp[-1] == 0                                  # deg h <= N-2
    finally:
        pk.free()


@pytest.mark.parametrize("curve,k", [("bn254", 10), ("bls12_381", 8)])
def test_mimc_chain_proof_trapdoor(ctx, curve, k):
    """BASELINE.json configs[0]: 2^10-constraint MiMC R1CS (S=102 -> 1020 constraints, N=2^10): the device proof
    equals the proof computed in the exponent from the toxic waste (oracle, no MSM/NTT involved except h)."""
    from ckb_zkp_amd.circuits import samples_for_domain
    S = samples_for_domain(k)
    inst = mimc_chain_instance(curve, S)
    params = groth16.generate_parameters(ctx, curve, inst, **TOXIC)
    pk = groth16.ProvingKey(ctx, params, inst)
    try:
        assert pk.domain_size == 1 << k
        c = params.curve
        z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
        h = codec.fr_from_mont(pk.witness_map(z), c)
        r_, s_ = 0xABCDEF0123456789, 0x9876543210FEDCBA
        out, inf = pk.prove_raw(z, codec.fr_to_mont([r_], c)[0], codec.fr_to_mont([s_], c)[0])
        proof = pk.decode_proof(out, inf)
        # expected, in the exponent
        t = params.toxic
        r = c.r
        ni = inst.num_inputs
        A = (t["alpha"] + sum(zi * ai for zi, ai in zip(inst.z, t["a"])) + r_ * t["delta"]) % r
        B = (t["beta"] + sum(zi * bi for zi, bi in zip(inst.z, t["b"])) + s_ * t["delta"]) % r
        L = sum(zi * li for zi, li in zip(inst.z[ni:], t["l"][ni:])) % r
        H = sum(hi * qi for hi, qi in zip(h, t["h"])) % r
        Cc = (s_ * A + r_ * B - r_ * s_ % r * t["delta"] + L + H) % r
        from oracle.pyref.curves import Group
        G1, G2 = Group(OC[curve], 1), Group(OC[curve], 2)
        assert proof.a == G1.mul(G1.gen, A)
        assert proof.b == G2.mul(G2.gen, B)
        assert proof.c == G1.mul(G1.gen, Cc)
        # h itself against the oracle's witness_map at this size (Python NTT of 2^k is still fast)
        cs = og.ConstraintSystem(OC[curve], True)
        og.MimcChain(OC[curve], inst.constants, inst.preimages).generate_constraints(cs)
        h_exp, _ = og.witness_map(cs)
        assert h == h_exp
    finally:
        pk.free()


def test_batch_prove_equals_sequential(ctx):
    """zkp_groth16_prove_batch_dev (two proofs in flight on two lanes) returns exactly the proofs of n blocking calls,
    for odd/even n, distinct witnesses and distinct (r, s) per proof."""
    import numpy as np
    from ckb_zkp_amd.circuits import samples_for_domain
    curve = "bn254"
    S = samples_for_domain(11)
    insts = [mimc_chain_instance(curve, S, seed=0xC0FFEE)]
    params = groth16.generate_parameters(ctx, curve, insts[0], **TOXIC)
    pk = groth16.ProvingKey(ctx, params, insts[0])
    c = params.curve
    try:
        # second witness for the same circuit: same constants, different preimages
        from ckb_zkp_amd.circuits import MimcChain
        from ckb_zkp_amd.r1cs import ConstraintSystem
        rnd = random.Random(8)
        cs = ConstraintSystem(curve, True)
        MimcChain(curve, insts[0].constants, [(rnd.randrange(c.r), rnd.randrange(c.r)) for _ in range(S)]).generate_constraints(cs)
        zs = [codec.fr_to_mont(insts[0].z, c).reshape(-1, 4), codec.fr_to_mont(cs.full_assignment(), c).reshape(-1, 4)]
        zd = [ctx.to_device(z) for z in zs]
        for n in (1, 2, 5, 9, 19):           # 8 lanes: 9 and 19 wrap around the lanes once and twice
            rs = [rnd.randrange(c.r) for _ in range(n)]
            ss = [rnd.randrange(c.r) for _ in range(n)]
            rm, sm = codec.fr_to_mont(rs, c), codec.fr_to_mont(ss, c)
            out, inf = pk.prove_batch_raw([zd[i % 2] for i in range(n)], rm, sm)
            for i in range(n):
                o1, i1 = pk.prove_raw(zd[i % 2], rm[i], sm[i], z_on_device=True)
                assert np.array_equal(out[i], o1) and np.array_equal(inf[i], i1), (n, i)
        for d in zd:
            ctx.dev_free(d)
    finally:
        pk.free()


@pytest.mark.parametrize("curve,k", [("bn254", 11), ("bls12_381", 9)])
def test_key_kept_as_given_proves_the_same(ctx, curve, k):
    """zkp_groth16_pk_upload_ex(ZKP_PK_KEEP_FORM) (one-shot callers should not pay the evaluation-form transforms): the key
    stays in the reference's form (pk_info: no evaluation-form H, no folded C), the proof bytes equal those of the transformed key and
    oracle/cpu's; an unknown flag bit is refused."""
    import ctypes as C
    import numpy as np
    from ckb_zkp_amd import _lib
    from ckb_zkp_amd.circuits import samples_for_domain
    from oracle import cpu_oracle
    inst = mimc_chain_instance(curve, samples_for_domain(k))
    params = groth16.generate_parameters(ctx, curve, inst, **TOXIC)
    c = params.curve
    z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
    rm, sm = codec.fr_to_mont([0xA5A5A5A5], c)[0], codec.fr_to_mont([0x5A5A5A5A5A], c)[0]
    pk = groth16.ProvingKey(ctx, params, inst)
    pk_raw = groth16.ProvingKey(ctx, params, inst, keep_form=True)
    try:
        assert pk.table_plan()["h_evaluation_form"] and pk.table_plan()["c_folded_into_l"]
        assert not pk_raw.table_plan()["h_evaluation_form"] and not pk_raw.table_plan()["c_folded_into_l"]
        o1, i1 = pk.prove_raw(z, rm, sm)
        o2, i2 = pk_raw.prove_raw(z, rm, sm)
        o3, i3, _ = cpu_oracle.groth16_prove(params, inst, z, rm, sm, threads=4)
        assert np.array_equal(o1, o2) and np.array_equal(i1, i2)
        assert np.array_equal(o2, o3) and np.array_equal(i2, i3)
        assert np.array_equal(pk_raw.witness_map(z), pk.witness_map(z))
        d, keep = groth16._fill_desc(params, inst)
        h = C.c_void_p()
        assert ctx.lib.zkp_groth16_pk_upload_ex(ctx.h, C.byref(d), 6, C.byref(h)) == -1          # ZKP_ERR_BAD_ARG
    finally:
        pk.free()
        pk_raw.free()


@pytest.mark.parametrize("curve,k,skew", [("bn254", 12, False), ("bls12_381", 11, False), ("bn254", 13, True)])
def test_device_keygen_equals_host_keygen(ctx, curve, k, skew):
    """generate_parameters(keygen="device") — the exponent vectors of large synthetic keys computed with the library's Fr vector
    kernels — returns the same Parameters (every query, flags, vk elements) and the same trapdoor exponents as the Python big-int
    path on the same instance; the sampled exponents re-derive from the CSR arrays (tests/util.spot_check_qap_exponents)."""
    import numpy as np
    from ckb_zkp_amd.circuits import samples_for_domain
    from tests.util import spot_check_qap_exponents
    if skew:
        from ckb_zkp_amd.circuits import boolean_mimc_instance
        inst = boolean_mimc_instance(curve, k)
    else:
        inst = mimc_chain_instance(curve, samples_for_domain(k))
    ph = groth16.generate_parameters(ctx, curve, inst, **TOXIC, keygen="host")
    pd = groth16.generate_parameters(ctx, curve, inst, **TOXIC, keygen="device")
    for name in ("alpha_g1", "beta_g1", "beta_g2", "gamma_g2", "delta_g1", "delta_g2"):
        assert np.array_equal(getattr(ph, name), getattr(pd, name)), name
    for name in ("gamma_abc_g1", "a_query", "b_g1_query", "b_g2_query", "h_query", "l_query"):
        (x1, i1), (x2, i2) = getattr(ph, name), getattr(pd, name)
        assert np.array_equal(x1, x2) and np.array_equal(i1, i2), name
    r = ph.curve.r
    for key in ("a", "b", "c", "l", "h"):
        assert list(pd.toxic[key]) == [v % r for v in ph.toxic[key]], key
    assert pd.toxic["zt"] == ph.toxic["zt"]
    spot_check_qap_exponents(pd, inst)


# configs[4]'s instance (2^24, BN254) is covered — single GPU and 8-way sharded — by tests/test_gpu_dist.py
from tests.util import TEST_FULL
_FULL = [("bn254", 20), ("bls12_381", 20), ("bls12_381", 22)] if TEST_FULL else [("bn254", 20)]


@pytest.mark.parametrize("curve,k", _FULL)
def test_full_size_proof_trapdoor_and_pipeline(ctx, curve, k):
    """BASELINE.json configs[1] at FULL size (1 048 570 constraints, domain 2^20, BN254), the same instance over
    BLS12-381, and configs[2] at full size (4 194 300 constraints, domain 2^22, BLS12-381): the device proof equals the proof computed in the exponent from the toxic waste (size-independent check,
    SURVEY §8(c).3), and the pipelined batch path (several proofs in flight on the context's lanes) returns the same
    proof as the blocking call."""
    from ckb_zkp_amd.circuits import samples_for_domain
    from oracle.pyref.curves import Group
    inst = mimc_chain_instance(curve, samples_for_domain(k))
    params = groth16.generate_parameters(ctx, curve, inst, **TOXIC)
    pk = groth16.ProvingKey(ctx, params, inst)
    try:
        assert pk.domain_size == 1 << k
        c = params.curve
        z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
        import numpy as np
        from oracle import cpu_oracle
        thr = cpu_oracle.hardware_threads()
        h_dev = pk.witness_map(z)
        # FULL-SIZE pin of h: the device's witness map equals the C++ restatement of
        # R1CStoQAP::witness_map (r1cs_to_qap.rs:113-172) limb for limb, not just "h[-1] == 0"
        r_, s_ = 0x1F2E3D4C5B6A7988, 0x8899AABBCCDDEEFF
        rm, sm = codec.fr_to_mont([r_], c)[0], codec.fr_to_mont([s_], c)[0]
        out, inf = pk.prove_raw(z, rm, sm)
        # ... and of the whole proof: device proof == the C++ restatement of create_proof (prover.rs:124-211) on all host threads
        # (one oracle pass hands back the proof and the quotient h it was made from)
        o_out, o_inf, _, o_h = cpu_oracle.groth16_prove(params, inst, z, rm, sm, threads=thr, want_h=True)
        assert np.array_equal(h_dev, o_h)
        assert not h_dev[-1].any()                             # deg h <= N - 2
        assert np.array_equal(out, o_out) and np.array_equal(inf, o_inf)
        proof = pk.decode_proof(out, inf)
        # the key's exponents (computed with the library's Fr kernels at this size) re-derived for a sample with Python integers,
        # then the proof in the exponent (inner products over the full assignment by oracle/cpu)
        from tests.util import spot_check_qap_exponents, trapdoor_proof_exponents
        spot_check_qap_exponents(params, inst)
        A, B, Cc = trapdoor_proof_exponents(params, inst, z, h_dev, r_, s_)
        G1, G2 = Group(OC[curve], 1), Group(OC[curve], 2)
        assert proof.a == G1.mul(G1.gen, A)
        assert proof.b == G2.mul(G2.gen, B)
        assert proof.c == G1.mul(G1.gen, Cc)
        # 7 proofs through the lanes, same (r, s): every one must be bit-identical to the blocking result
        zd = ctx.to_device(z)
        import numpy as np
        outs, infs = pk.prove_batch_raw([zd] * 7, np.stack([rm] * 7), np.stack([sm] * 7))
        for i in range(7):
            assert np.array_equal(outs[i], out) and np.array_equal(infs[i], inf), i
        ctx.dev_free(zd)
    finally:
        pk.free()


def test_graph_replay_equals_eager():
    """ZKP_GRAPH=1: the third and later proofs on a lane replay a captured hipGraph; they must equal the eager proofs
    (own process: the switch is read once per process)."""
    code = r'''
import numpy as np, random
from ckb_zkp_amd import codec, groth16
from ckb_zkp_amd.api import Context
from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain
TOXIC = dict(alpha=11, beta=13, gamma=17, delta=19, tau=23)
ctx = Context(0)
inst = mimc_chain_instance("bn254", samples_for_domain(12))
params = groth16.generate_parameters(ctx, "bn254", inst, **TOXIC)
pk = groth16.ProvingKey(ctx, params, inst)
c = params.curve
z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
zd = ctx.to_device(z)
rnd = random.Random(4)
n = 7
rs = codec.fr_to_mont([rnd.randrange(c.r) for _ in range(n)], c)
ss = codec.fr_to_mont([rnd.randrange(c.r) for _ in range(n)], c)
outs, infs = pk.prove_batch_raw([zd] * n, rs, ss)
print("PROOFS", outs.tobytes().hex(), infs.tobytes().hex())
'''
    res = {g: _proofs_in_subprocess(code, ZKP_GRAPH=g, ZKP_LANES="2") for g in ("0", "1")}
    assert res["0"] == res["1"]


def _proofs_in_subprocess(code, **env_extra):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, **env_extra)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    return [l for l in out.stdout.splitlines() if l.startswith("PROOFS")][0]


_SWITCH_CODE = r'''
import numpy as np, random
from ckb_zkp_amd import codec, groth16
from ckb_zkp_amd.api import Context
from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain
TOXIC = dict(alpha=11, beta=13, gamma=17, delta=19, tau=23)
ctx = Context(0)
inst = mimc_chain_instance("bn254", samples_for_domain(12))
params = groth16.generate_parameters(ctx, "bn254", inst, **TOXIC)
pk = groth16.ProvingKey(ctx, params, inst)
c = params.curve
z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
zd = ctx.to_device(z)
rnd = random.Random(9)
n = 3
rs = codec.fr_to_mont([rnd.randrange(c.r) for _ in range(n)], c)
ss = codec.fr_to_mont([rnd.randrange(c.r) for _ in range(n)], c)
outs, infs = pk.prove_batch_raw([zd] * n, rs, ss)
one, inf1 = pk.prove_raw(z, rs[0], ss[0])
assert np.array_equal(one, outs[0]) and np.array_equal(inf1, infs[0])
print("PROOFS", outs.tobytes().hex(), infs.tobytes().hex())
'''


def test_schedule_switches_do_not_change_the_proof():
    """Every scheduling shortcut of the prover — bucket chaining (H accumulates into L's buckets, one reduction for l' + h_acc),
    the shared level-1 pass, shared sorts, the staged scatter, the three-stream plan, window groups, the unsaturated NTT pass and its
    full-size tables / fused chains, the evaluation-form key (transformed H query, C folded into the L query), the host-side
    into_affine — is an optimisation only: proofs are
    byte-identical with each one switched off (own processes: the switches are read once per process)."""
    base = _proofs_in_subprocess(_SWITCH_CODE)
    for sw in ({"ZKP_CHAIN_LH": "0"}, {"ZKP_SHARE_L1": "0"}, {"ZKP_SHARE_B_SORT": "0", "ZKP_SHARE_AL_SORT": "0"},
               {"ZKP_SORT_STAGED": "0"}, {"ZKP_SINGLE_STREAM": "1"}, {"ZKP_LATENCY_PLAN": "0"}, {"ZKP_TABLE_K": "2"},
               {"ZKP_NTT_V2": "0"}, {"ZKP_NTT_FULL": "0"}, {"ZKP_NTT_FUSE": "0"},
               # the key in coefficient form (no transformed H query, C not folded into L), H alone transformed, host / device into_affine
               {"ZKP_H_LAGRANGE": "0"}, {"ZKP_C_FOLD": "0"}, {"ZKP_HOST_AFFINE": "0"},
               # every eighth accumulate task through the exact (redo) kernel: on top of chained buckets, and without chaining
               {"ZKP_DEBUG_FORCE_REDO": "1"}, {"ZKP_DEBUG_FORCE_REDO": "1", "ZKP_CHAIN_LH": "0"}):
        assert _proofs_in_subprocess(_SWITCH_CODE, **sw) == base, sw



@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_create_random_proof_and_no_zk_pass_the_reference_verifier(ctx, curve):
    """The reference's own round trip (groth16/tests/mini.rs:80-96): create_random_proof -> verify_proof == true, and
    the same for create_proof_no_zk (prover.rs:113-122); a wrong public input and a tampered proof are rejected.
    The verifier is the oracle's from-scratch pairing check of groth16/src/verifier.rs:18-44."""
    from oracle.pyref.pairing import verify_proof
    from oracle.pyref.curves import Group
    params = groth16.generate_parameters(ctx, curve, Mini(num=10), **TOXIC, g1_k=7, g2_k=9)
    vk = _oracle_params(curve, og.MiniCircuit(num=10), 7, 9)           # same trapdoor -> same verifying key
    c = params.curve
    assert codec.g1_from_mont(*params.gamma_abc_g1, c) == vk.gamma_abc_g1
    pk = groth16.ProvingKey(ctx, params, Mini(num=10))
    try:
        rng = random.Random(2024)
        p1 = groth16.create_random_proof(pk, Mini(2, 3, 10, 10), rng)
        p2 = groth16.create_random_proof(pk, Mini(2, 3, 10, 10), rng)
        p0 = groth16.create_proof_no_zk(pk, Mini(2, 3, 10, 10))
        assert (p1.a, p1.b, p1.c) != (p2.a, p2.b, p2.c)                # fresh (r, s) every time
        for p in (p1, p2, p0):
            assert verify_proof(OC[curve], vk, og.Proof(p.a, p.b, p.c), [10])
        assert not verify_proof(OC[curve], vk, og.Proof(p1.a, p1.b, p1.c), [11])
        G1 = Group(OC[curve], 1)
        assert not verify_proof(OC[curve], vk, og.Proof(G1.add(p1.a, G1.gen), p1.b, p1.c), [10])
        # no_zk == create_proof with (r, s) = (0, 0)
        q = groth16.create_proof(pk, Mini(2, 3, 10, 10), 0, 0)
        assert (p0.a, p0.b, p0.c) == (q.a, q.b, q.c)
        # OS randomness path (rng=None)
        p3 = groth16.create_random_proof(pk, Mini(2, 3, 10, 10))
        assert verify_proof(OC[curve], vk, og.Proof(p3.a, p3.b, p3.c), [10])
    finally:
        pk.free()


class _DenseChain:
    """t_i = (x_i + x_{i+1} + t_{i-1}) * x_i: every auxiliary variable but the last product occurs in an A row, so the A and L
    queries have (almost) the same identity pattern — the case in which L reuses A's bucket sort (groth16.hip share_al_sort)."""

    def __init__(self, curve, xs):
        self.r = OC[curve].r
        self.xs = xs

    def generate_constraints(self, cs):
        r = self.r
        out = cs.alloc_input(lambda: 7)
        xv = [cs.alloc(lambda v=v: v) for v in self.xs]
        cs.enforce(lambda lc: lc + out, lambda lc: lc + cs.one(), lambda lc: lc + (7, cs.one()))
        t_var, t_val = None, 0
        for i in range(len(self.xs) - 1):
            a_val = (self.xs[i] + self.xs[i + 1] + t_val) % r
            new_val = a_val * self.xs[i] % r
            new_var = cs.alloc(lambda v=new_val: v)
            prev = t_var
            cs.enforce(lambda lc: (lc + xv[i] + xv[i + 1]) if prev is None else (lc + xv[i] + xv[i + 1] + prev),
                       lambda lc: lc + xv[i], lambda lc: lc + new_var)
            t_var, t_val = new_var, new_val


def test_l_reuses_a_sort_when_identity_patterns_agree(ctx, capfd, monkeypatch):
    """Proof == trapdoor-expected proof on a circuit whose A and L queries differ in one identity base (sort shared)."""
    from ckb_zkp_amd.r1cs import ConstraintSystem
    from oracle.pyref.curves import Group
    curve = "bn254"
    rnd = random.Random(21)
    circ = _DenseChain(curve, [rnd.randrange(OC[curve].r) for _ in range(300)])
    params = groth16.generate_parameters(ctx, curve, circ, **TOXIC)
    monkeypatch.setenv("ZKP_DEBUG_MSM", "0")
    pk = groth16.ProvingKey(ctx, params, circ)
    try:
        assert "A/L sort sharing: 1" in capfd.readouterr().err
        c = params.curve
        cs = ConstraintSystem(curve, True)
        circ.generate_constraints(cs)
        zi = cs.full_assignment()
        z = codec.fr_to_mont(zi, c).reshape(-1, 4)
        h = codec.fr_from_mont(pk.witness_map(z), c)
        t, r, ni = params.toxic, c.r, 2
        for r_, s_ in ((0x1234567, 0x7654321), (0, 0)):
            out, inf = pk.prove_raw(z, codec.fr_to_mont([r_], c)[0], codec.fr_to_mont([s_], c)[0])
            proof = pk.decode_proof(out, inf)
            A = (t["alpha"] + sum(x * y for x, y in zip(zi, t["a"])) + r_ * t["delta"]) % r
            B = (t["beta"] + sum(x * y for x, y in zip(zi, t["b"])) + s_ * t["delta"]) % r
            L = sum(x * y for x, y in zip(zi[ni:], t["l"][ni:])) % r
            H = sum(x * y for x, y in zip(h, t["h"])) % r
            Cc = (s_ * A + r_ * B - r_ * s_ % r * t["delta"] + L + H) % r
            G1, G2 = Group(OC[curve], 1), Group(OC[curve], 2)
            assert proof.a == G1.mul(G1.gen, A)
            assert proof.b == G2.mul(G2.gen, B)
            assert proof.c == G1.mul(G1.gen, Cc)
    finally:
        pk.free()


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_identity_proof_points_through_the_host_tail(ctx, curve):
    """The three proof points leave the device as XYZZ and become affine on the host (host_field.hpp, one inversion for all three):
    the identity must come out as ark's (0, 0) + flag, alone and next to ordinary points.  (i) a key whose G2 side is the identity
    everywhere: B = O while A and C are those of the ordinary key (C does not touch G2, prover.rs:192-210); (ii) an all-identity
    key: A = B = C = O.  ark's `Parameters::deserialize` accepts such keys (a useless key, not a malformed one)."""
    import dataclasses

    import numpy as np

    S = 20
    inst = mimc_chain_instance(curve, S)
    params = groth16.generate_parameters(ctx, curve, inst, **TOXIC)
    c = params.curve
    z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
    rm, sm = codec.fr_to_mont([0x1234567], c)[0], codec.fr_to_mont([0x7654321], c)[0]

    def ident(q):
        return (np.zeros_like(q[0]), np.ones_like(q[1]))

    pk = groth16.ProvingKey(ctx, params, inst)
    try:
        out0, inf0 = pk.prove_raw(z, rm, sm)
        p0 = pk.decode_proof(out0, inf0)
    finally:
        pk.free()
    assert list(inf0) == [0, 0, 0]
    g2_off = dataclasses.replace(params, beta_g2=np.zeros_like(params.beta_g2), delta_g2=np.zeros_like(params.delta_g2),
                                 b_g2_query=ident(params.b_g2_query))
    pk = groth16.ProvingKey(ctx, g2_off, inst)
    try:
        out, inf = pk.prove_raw(z, rm, sm)
        p = pk.decode_proof(out, inf)
    finally:
        pk.free()
    assert list(inf) == [0, 1, 0] and p.a == p0.a and p.c == p0.c and p.b is None
    fq = c.fq_limbs
    assert not out[2 * fq:6 * fq].any()                                   # the identity is written as zeros
    all_off = dataclasses.replace(g2_off, alpha_g1=np.zeros_like(params.alpha_g1), beta_g1=np.zeros_like(params.beta_g1),
                                  delta_g1=np.zeros_like(params.delta_g1), a_query=ident(params.a_query),
                                  b_g1_query=ident(params.b_g1_query), h_query=ident(params.h_query),
                                  l_query=ident(params.l_query))
    pk = groth16.ProvingKey(ctx, all_off, inst)
    try:
        out, inf = pk.prove_raw(z, rm, sm)
        outs, infs = pk.prove_batch_raw([ctx.to_device(z)] * 3, np.stack([rm] * 3), np.stack([sm] * 3))
    finally:
        pk.free()
    assert list(inf) == [1, 1, 1] and not out.any()
    assert not outs.any() and infs.tolist() == [[1, 1, 1]] * 3
