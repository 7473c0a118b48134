"""Seeded differential fuzz of the two primitives the path is made of: random shapes, random identity
patterns, adversarial scalar classes — HIP through the C ABI against oracle/cpu (the C++ restatement of ark-ec's window-parallel
Pippenger and ark-poly's radix-2 transforms), both as group elements / field elements, bit for bit.  Complements the hand-picked
edge sets of test_gpu_msm.py / test_gpu_ntt.py; 480 MSM cases (+ 120 through the variable-base entry point) and 480 transforms, ≈ 25 s on the GPU box."""
import os

import numpy as np
import pytest

from ckb_zkp_amd import codec
from ckb_zkp_amd.api import NTT_COSET_FFT, NTT_COSET_IFFT, NTT_FFT, NTT_IFFT
from ckb_zkp_amd.params import get_curve
from oracle import cpu_oracle
from oracle.pyref.curves import Group
from tests.util import OC, jac_limbs_to_affine_oracle, to_abi_points

pytestmark = pytest.mark.gpu
SEED = int(os.environ.get("ZKP_FUZZ_SEED", "0"))        # 0 = the committed cases; any other value: a fresh set (soak runs)


def _scalars(rng, c, n, kind):
    """canonical scalars < r as (n, 4) uint64.  kinds stress: uniform, booleans, small, near r, one hot window, all equal."""
    r = c.r
    if kind == "uniform":
        v = [int.from_bytes(rng.bytes(40), "little") % r for _ in range(n)]
    elif kind == "bool":
        v = [int(x) for x in rng.integers(0, 2, n)]
    elif kind == "small":
        v = [int(x) for x in rng.integers(0, 1 << 20, n)]
    elif kind == "near_r":
        v = [r - 1 - int(x) for x in rng.integers(0, 1 << 16, n)]
    elif kind == "one_window":                                   # a single non-zero 20-bit window at a random position per scalar
        v = [(int(x) << int(s)) % r for x, s in zip(rng.integers(1, 1 << 20, n), rng.integers(0, 235, n))]
    else:                                                        # "equal": every scalar the same full-width value
        k = int.from_bytes(rng.bytes(40), "little") % r
        v = [k] * n
    return codec.fr_canonical(v, c).reshape(-1, 4) if n else np.zeros((0, 4), dtype=np.uint64)


@pytest.mark.parametrize("curve,group,cases", [("bn254", 1, 240), ("bn254", 2, 96), ("bls12_381", 1, 96), ("bls12_381", 2, 48)])
def test_msm_fuzz_against_cpu_port(ctx, curve, group, cases):
    c = get_curve(curve)
    rng = np.random.default_rng(0xF00D + 17 * group + c.cid + 1000003 * SEED)
    w = 2 * c.fq_limbs * group
    nmax = 20000
    # one pool of bases k_i * G built on the device (checked against the oracle elsewhere), with duplicates and a few P / -P pairs
    G = Group(OC[curve], group)
    gen, _ = to_abi_points(curve, group, [G.gen])
    pool, _ = ctx.fixed_base_mul(c, group, gen, _scalars(rng, c, nmax, "uniform"))
    pool[7] = pool[3]                                            # duplicates: the doubling branch inside a bucket
    f = c.fq_limbs
    y = codec.limbs_to_ints(pool[11, w // 2:].reshape(-1, f))    # pool[12] = -pool[11]: cancellation inside a bucket
    pool[12, :w // 2] = pool[11, :w // 2]
    pool[12, w // 2:] = codec.ints_to_limbs([(c.q - v) % c.q for v in y], f).reshape(-1)
    kinds = ["uniform", "bool", "small", "near_r", "one_window", "equal"]
    for case in range(cases):
        n = int(rng.choice([0, 1, 2, 3, 63, 64, 65, 255, 257, 1000, int(rng.integers(1, nmax))]))
        off = int(rng.integers(0, nmax - n + 1))
        nb = n + int(rng.integers(0, 5)) if rng.random() < 0.3 else n          # more bases than scalars: ark's min(len) truncation
        nb = min(nb, nmax - off)
        xy = pool[off:off + nb].copy()
        inf = (rng.random(nb) < rng.choice([0.0, 0.05, 0.6, 1.0])).astype(np.uint8)
        xy[inf != 0] = 0
        kind = kinds[case % len(kinds)]
        ns = n + (int(rng.integers(0, 5)) if rng.random() < 0.2 else 0)        # ... or more scalars than bases
        sc = _scalars(rng, c, ns, kind)
        exp = cpu_oracle.msm(c.cid, group, xy, inf, sc, threads=4)
        bases = ctx.upload_bases(c, group, xy, inf)
        try:
            got = bases.msm(sc)
        finally:
            bases.free()
        tag = (curve, group, case, n, nb, ns, kind, int(inf.sum()))
        assert jac_limbs_to_affine_oracle(curve, group, got) == jac_limbs_to_affine_oracle(curve, group, exp), tag
        if case % 4 == 0:                                                      # the true variable-base entry point (nothing resident)
            got_v = ctx.msm_var(c, group, xy, inf, sc)
            assert jac_limbs_to_affine_oracle(curve, group, got_v) == jac_limbs_to_affine_oracle(curve, group, exp), ("var",) + tag


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_ntt_fuzz_against_cpu_port(ctx, curve):
    c = get_curve(curve)
    rng = np.random.default_rng(0xBEEF + c.cid + 1000003 * SEED)
    for case in range(120):
        k = int(rng.integers(0, 19))
        n = 1 << k
        kind = case % 4
        if kind == 0:
            v = [int.from_bytes(rng.bytes(40), "little") % c.r for _ in range(n)]
        elif kind == 1:
            v = [0] * n                                                        # the zero vector stays zero
        elif kind == 2:
            v = [c.r - 1] * n
        else:
            v = [0] * n
            v[int(rng.integers(0, n))] = 1                                     # a unit vector -> a geometric sequence
        x = codec.fr_to_mont(v, c).reshape(-1, 4)
        for op in (NTT_FFT, NTT_IFFT, NTT_COSET_FFT, NTT_COSET_IFFT):
            assert np.array_equal(ctx.ntt(c, x, op), cpu_oracle.ntt(c.cid, x, op, threads=4)), (curve, case, k, kind, op)


class _RandomCircuit:
    """Random R1CS through the reference's ConstraintSystem interface (r1cs/src/constraint_system.rs:10-93): rows with 0-4 terms,
    coefficients from {1, -1, small, full width}, repeated variables inside a row, public inputs beyond the constant, and a constraint
    count that lands below / on / above a power of two.  Most rows are satisfied (c = a fresh variable holding a*b); a few are not —
    the prover's formulas (prover.rs:124-211, r1cs_to_qap.rs:113-172) are defined either way and both sides must agree."""

    def __init__(self, curve, seed, n_inputs, n_constraints, unsatisfied=True):
        self.c, self.seed, self.ni, self.nc, self.unsat = get_curve(curve), seed, n_inputs, n_constraints, unsatisfied

    def generate_constraints(self, cs):
        import random
        rnd = random.Random(self.seed)
        r = self.c.r
        vals = {cs.one(): 1}
        vs = [cs.one()]
        for _ in range(self.ni):
            v = rnd.randrange(r)
            x = cs.alloc_input(lambda v=v: v)
            vals[x] = v
            vs.append(x)
        for _ in range(3):
            v = rnd.choice([0, 1, rnd.randrange(r)])
            x = cs.alloc(lambda v=v: v)
            vals[x] = v
            vs.append(x)

        def coeff():
            return rnd.choice([1, 1, r - 1, rnd.randrange(1, 1000), rnd.randrange(r)])

        def row():
            return [(coeff(), rnd.choice(vs)) for _ in range(rnd.choice([0, 1, 1, 2, 2, 3, 4]))]

        def value(terms):
            return sum(k * vals[v] for k, v in terms) % r

        def lc_of(terms):
            def f(lc):
                for k, v in terms:
                    lc = lc + (k, v)
                return lc
            return f

        for i in range(self.nc):
            ta, tb = row(), row()
            prod = value(ta) * value(tb) % r
            kind = rnd.random() * (1.0 if self.unsat else 0.9)
            if kind < 0.75:                                   # c = one fresh variable (unit coefficient: the skip-the-product path)
                x = cs.alloc(lambda v=prod: v)
                vals[x] = prod
                vs.append(x)
                tc = [(1, x)]
            elif kind < 0.9:                                  # c = k * fresh + existing terms, still satisfied
                k = rnd.randrange(1, r)
                rest = row()
                want = (prod - value(rest)) * pow(k, -1, r) % r
                x = cs.alloc(lambda v=want: v)
                vals[x] = want
                vs.append(x)
                tc = [(k, x)] + rest
            else:                                             # arbitrary c: not satisfied
                tc = row()
            cs.enforce(lc_of(ta), lc_of(tb), lc_of(tc))


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_groth16_fuzz_against_cpu_port(ctx, curve):
    """Random circuits: device proof and witness map == oracle/cpu (same key, witness, r, s), bit for bit."""
    from ckb_zkp_amd import groth16
    from ckb_zkp_amd.r1cs import ConstraintSystem, R1csInstance
    c = get_curve(curve)
    rng = np.random.default_rng(0xC1C + c.cid + 1000003 * SEED)
    toxic = dict(alpha=0x1234567, beta=0x89ABCDE, gamma=0xF012345, delta=0x6789ABC, tau=0xDEF0123456789)
    shapes = [(1, 1), (0, 1), (2, 2), (1, 3), (3, 4), (0, 7), (1, 8), (2, 13), (1, 15), (1, 16), (4, 17), (1, 30), (0, 31), (2, 64),
              (1, 100), (3, 127), (0, 128), (1, 129), (2, 250), (1, 511), (1, 600)]
    for case, (ni, nc) in enumerate(shapes):
        cs = ConstraintSystem(c, True)
        _RandomCircuit(curve, 1000 + case + 7919 * SEED, ni, nc).generate_constraints(cs)
        inst = R1csInstance.from_cs(cs)
        params = groth16.generate_parameters(ctx, c, inst, **toxic)
        pk = groth16.ProvingKey(ctx, params, inst)
        try:
            z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
            assert np.array_equal(pk.witness_map(z), cpu_oracle.witness_map(params, inst, z, threads=4)), (curve, case, ni, nc)
            for _ in range(2):
                rm = codec.fr_to_mont([int.from_bytes(rng.bytes(40), "little") % c.r], c)[0]
                sm = codec.fr_to_mont([int.from_bytes(rng.bytes(40), "little") % c.r], c)[0]
                out, inf = pk.prove_raw(z, rm, sm)
                o_out, o_inf, _ = cpu_oracle.groth16_prove(params, inst, z, rm, sm, threads=4)
                assert np.array_equal(out, o_out) and np.array_equal(inf, o_inf), (curve, case, ni, nc)
        finally:
            pk.free()


class _DenseColumnCircuit:
    """x_i * x_i_inv = ONE-ish rows whose C side all reference the SAME two variables (the constant one with unit / -1 / general
    coefficients, and one witness variable): dense columns of C, the shape ADVICE r4 flagged for the C fold at key upload."""

    def __init__(self, curve, seed, rows):
        self.c, self.seed, self.rows = get_curve(curve), seed, rows

    def generate_constraints(self, cs):
        import random
        rnd = random.Random(self.seed)
        r = self.c.r
        hv = rnd.randrange(1, r)
        hub = cs.alloc(lambda: hv)
        for i in range(self.rows):
            a = rnd.randrange(1, r)
            k1 = rnd.choice([1, r - 1, rnd.randrange(1, r)])
            k2 = rnd.choice([0, 1, rnd.randrange(1, r)])
            # a * b = k1 * ONE + k2 * hub  ->  b = (k1 + k2 hub) / a
            bval = (k1 + k2 * hv) * pow(a, -1, r) % r
            xa = cs.alloc(lambda v=a: v)
            xb = cs.alloc(lambda v=bval: v)
            cs.enforce(lambda lc, xa=xa: lc + (1, xa), lambda lc, xb=xb: lc + (1, xb),
                       lambda lc, k1=k1, k2=k2: (lc + (k1, cs.one()) + (k2, hub)) if k2 else (lc + (k1, cs.one())))


class _ManyHubsCircuit:
    """`hubs` witness variables, each on the C side of `per_hub` rows with a general coefficient (cost 380 each: a medium-dense
    column), next to the constant-one column that every row touches: the shape ADVICE r5 flagged — with a fixed threshold every hub
    above it becomes a serial MSM of its own at key upload."""

    def __init__(self, curve, seed, hubs, per_hub):
        self.c, self.seed, self.hubs, self.per_hub = get_curve(curve), seed, hubs, per_hub

    def generate_constraints(self, cs):
        import random
        rnd = random.Random(self.seed)
        r = self.c.r
        for _ in range(self.hubs):
            hv = rnd.randrange(1, r)
            hub = cs.alloc(lambda v=hv: v)
            for _ in range(self.per_hub):
                a, k1, k2 = rnd.randrange(1, r), rnd.randrange(1, r), rnd.randrange(2, r - 1)
                bval = (k1 + k2 * hv) * pow(a, -1, r) % r
                xa = cs.alloc(lambda v=a: v)
                xb = cs.alloc(lambda v=bval: v)
                cs.enforce(lambda lc, xa=xa: lc + (1, xa), lambda lc, xb=xb: lc + (1, xb),
                           lambda lc, k1=k1, k2=k2, hub=hub: lc + (k1, cs.one()) + (k2, hub))


@pytest.mark.parametrize("hubs,per_hub", [(40, 150), (12, 100)])
def test_groth16_key_fold_cut_is_chosen_per_key(ctx, hubs, per_hub):
    """The heavy-column cut of fold_c_into_l, calibrated by this test in round 6: a lone lane's point operation takes 8-16 µs,
    a heavy column's MSM 1.7-3 ms ≈ 150 operations (first measurements: 40 x 57 000: kernel 0.54 s, 41 MSMs 0.13 s; 300 x 53 200: kernel
    0.88 s, 301 MSMs 0.52 s).  40 columns of cost 57 000 + the constant-one column: the per-key cut and the fixed 50 000 of round 5 agree
    (all MSMs).  12 columns of cost 38 000: the fixed threshold keeps them in the kernel (a 0.3-0.6 s chain), the per-key cut sends them
    through twelve MSMs.  Same key either way — both proofs equal oracle/cpu's — and the default must not be the slower one."""
    import time
    from ckb_zkp_amd import groth16
    from ckb_zkp_amd.api import Context
    from ckb_zkp_amd.r1cs import ConstraintSystem, R1csInstance
    c = get_curve("bn254")
    cs = ConstraintSystem(c, True)
    _ManyHubsCircuit("bn254", 77, hubs, per_hub).generate_constraints(cs)
    inst = R1csInstance.from_cs(cs)
    params = groth16.generate_parameters(ctx, c, inst, alpha=0x7654321, beta=0x1ABCDE, gamma=0xF0145, delta=0x67ABC, tau=0xDEF01236789)
    z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
    rm, sm = codec.fr_to_mont([0x1234567], c)[0], codec.fr_to_mont([0x89ABCDE], c)[0]
    o_out, o_inf, _ = cpu_oracle.groth16_prove(params, inst, z, rm, sm, threads=8)
    secs = {}
    for cost in (0, 50000):
        own = Context(ctx.device, dict(c_fold_heavy_cost=cost) if cost else None)
        try:
            groth16.ProvingKey(own, params, inst).free()           # first upload of a context allocates its scratch
            t = time.perf_counter()
            pk = groth16.ProvingKey(own, params, inst)
            secs[cost] = time.perf_counter() - t
            assert pk.table_plan()["c_folded_into_l"]
            out, inf = pk.prove_raw(z, rm, sm)
            assert np.array_equal(out, o_out) and np.array_equal(inf, o_inf), cost
            pk.free()
        finally:
            own.close()
    print(f"key upload ({hubs} x {per_hub}): per-key cut {secs[0]:.3f} s, fixed 50000 {secs[50000]:.3f} s")
    assert secs[0] <= secs[50000] * 1.25 + 0.05


@pytest.mark.parametrize("curve,rows,cost", [("bn254", 300, 3), ("bls12_381", 90, 3), ("bn254", 700, 50000), ("bn254", 700, 0)])
def test_groth16_dense_c_column_takes_the_heavy_fold_path(ctx, curve, rows, cost):
    """fold_c_into_l (csrc/groth16.hip): columns of C above zkp_ctx_config.c_fold_heavy_cost leave the one-lane-per-variable kernel and
    take one variable-base MSM each (gathered one column at a time); the proof and witness map still equal oracle/cpu's.  cost = 3
    forces nearly every column through the MSM path; 50000 with 700 rows sends exactly the two dense columns (constant one + hub)
    there; 0 = the default, a cut chosen per key (longest kernel chain + 150 per heavy column minimised: the same two columns here).
    The key is uploaded through a context of its own with that configuration; the session context generates the parameters."""
    from ckb_zkp_amd import groth16
    from ckb_zkp_amd.api import Context
    from ckb_zkp_amd.r1cs import ConstraintSystem, R1csInstance
    own = Context(ctx.device, dict(c_fold_heavy_cost=cost) if cost else None)
    assert own.config()["c_fold_heavy_cost"] == cost
    c = get_curve(curve)
    toxic = dict(alpha=0x7654321, beta=0x1ABCDE, gamma=0xF0145, delta=0x67ABC, tau=0xDEF01236789)
    cs = ConstraintSystem(c, True)
    _DenseColumnCircuit(curve, 4242 + rows, rows).generate_constraints(cs)
    inst = R1csInstance.from_cs(cs)
    params = groth16.generate_parameters(ctx, c, inst, **toxic)
    pk = groth16.ProvingKey(own, params, inst)
    try:
        assert pk.table_plan()["c_folded_into_l"]
        z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
        rng = np.random.default_rng(rows)
        for _ in range(2):
            rm = codec.fr_to_mont([int.from_bytes(rng.bytes(40), "little") % c.r], c)[0]
            sm = codec.fr_to_mont([int.from_bytes(rng.bytes(40), "little") % c.r], c)[0]
            out, inf = pk.prove_raw(z, rm, sm)
            o_out, o_inf, _ = cpu_oracle.groth16_prove(params, inst, z, rm, sm, threads=4)
            assert np.array_equal(out, o_out) and np.array_equal(inf, o_inf)
    finally:
        pk.free()
        own.close()


@pytest.mark.parametrize("curve,seed,ni,nc", [("bn254", 1, 1, 5), ("bn254", 2, 0, 9), ("bls12_381", 3, 1, 6), ("bn254", 4, 3, 14), ("bls12_381", 5, 0, 3), ("bn254", 6, 1, 30), ("bls12_381", 7, 3, 21)])
def test_marlin_fuzz_against_oracle(ctx, curve, seed, ni, nc):
    """Random (satisfied) circuits through zkp_marlin_index_upload / _index_commit / _prove against oracle/pyref's create_random_proof:
    index commitments, transcript-derived challenges, commitments, 21 evaluations, both opening proofs; the oracle's verifier accepts.
    (1 + ni must be a power of two: the AHP indexes the public inputs by a subdomain of H, marlin/src/ahp/prover.rs:86-147.)"""
    import random

    from ckb_zkp_amd import kzg10
    from ckb_zkp_amd import marlin as marlin_native
    from ckb_zkp_amd.r1cs import ConstraintSystem, R1csInstance
    from oracle.pyref import kzg10 as okzg
    from oracle.pyref import marlin as om
    c = get_curve(curve)
    circ = _RandomCircuit(curve, 7000 + seed, ni, nc, unsatisfied=False)
    cs = ConstraintSystem(c, True)
    circ.generate_constraints(cs)
    inst = R1csInstance.from_cs(cs)
    oidx = om.index(OC[curve], circ)
    nidx = marlin_native.NativeIndex(ctx, inst)
    assert (nidx.xs, nidx.hs, nidx.ks, nidx.bs, nidx.max_degree, nidx.num_non_zeros) == \
        (oidx["dx"].size, oidx["dh"].size, oidx["dk"].size, oidx["db"].size, oidx["max_degree"], oidx["num_non_zeros"])
    beta_srs = 0x2468ACE13579BDF + seed
    pp = okzg.setup(OC[curve], nidx.max_degree, beta_srs)
    ck = kzg10.setup(ctx, curve, nidx.max_degree, beta_srs)
    try:
        ic = om.index_commitments(oidx, pp)
        assert nidx.commit_index(ck) == ic
        rnd = random.Random(seed)
        R = dict(w=[rnd.randrange(c.r)], z_a=[rnd.randrange(c.r)], z_b=[rnd.randrange(c.r)],
                 mask=[rnd.randrange(c.r) for _ in range(3 * nidx.hs)],
                 blind={l: [rnd.randrange(c.r), rnd.randrange(c.r)] for l in ("w", "z_a", "z_b", "g_1")},
                 blind_shifted={"g_1": [rnd.randrange(c.r), rnd.randrange(c.r)]})
        x, w = inst.z[:inst.num_inputs], inst.z[inst.num_inputs:]
        ivk = om.index_verifier_key(oidx, pp, ic)
        p = marlin_native.prove_native(ctx, nidx, ck, ivk, x, w, R)
        o = om.create_random_proof(oidx, pp, ic, circ, R)
        assert p["challenges"] == o["challenges"]
        assert p["commitments"] == o["commitments"] and p["evaluations"] == o["evaluations"]
        assert p["opening_proofs"] == o["opening_proofs"]
        wire = dict(commitments=p["commitments"], evaluations=p["evaluations"], opening_proofs=p["opening_proofs"])
        assert om.verify_random_proof(oidx, pp, ic, wire, x[1:])
    finally:
        nidx.free()
        ck.powers_of_g.free()
        ck.powers_of_gamma_g.free()
