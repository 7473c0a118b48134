"""MSM parity: HIP Pippenger (C ABI) vs the oracle.  Replaces ark-ec VariableBaseMSM::multi_scalar_mul
(/root/reference/groth16/src/prover.rs:187,190,220)."""
import random

import numpy as np
import pytest

from ckb_zkp_amd import codec
from ckb_zkp_amd.params import get_curve
from oracle.pyref.curves import Group
from tests.util import OC, jac_limbs_to_affine_oracle, jac_to_affine, random_points, to_abi_points

pytestmark = pytest.mark.gpu
CFG = [("bn254", 1), ("bn254", 2), ("bls12_381", 1), ("bls12_381", 2)]


@pytest.mark.parametrize("curve,group", CFG)
def test_msm_small_edge_cases(ctx, curve, group):
    """zero/one/r-1/2^k scalars, identity bases, duplicate bases (doubling branch), P and -P, n < len(bases)."""
    c = get_curve(curve)
    G = Group(OC[curve], group)
    rnd = random.Random(7)
    pts = random_points(curve, group, 24, seed=11 + group)
    pts[3] = None                      # identity base
    pts[5] = pts[4]                    # duplicate -> doubling inside a bucket
    pts[7] = G.neg(pts[6])             # P, -P
    ks = [rnd.randrange(c.r) for _ in range(24)]
    ks[0], ks[1], ks[2], ks[8], ks[9] = 0, 1, c.r - 1, 2, 1 << 253
    ks[5] = ks[4]                      # same scalar for the duplicate pair -> same buckets
    ks[7] = ks[6]                      # k*P + k*(-P) = 0 inside the buckets
    xy, inf = to_abi_points(curve, group, pts)
    bases = ctx.upload_bases(c, group, xy, inf)
    try:
        for n, off in ((24, 0), (17, 3), (1, 0), (0, 0), (30, 0), (5, 19)):
            sc = codec.fr_canonical(ks[:n], c).reshape(-1, 4)
            out = bases.msm(sc, off)
            m = min(n, 24 - off)
            exp = G.msm_naive(pts[off:off + m], ks[:m])
            assert jac_limbs_to_affine_oracle(curve, group, out) == exp, (n, off)
            assert jac_to_affine(ctx, curve, group, out) == exp
    finally:
        bases.free()


@pytest.mark.parametrize("curve,group", CFG)
def test_msm_all_equal_scalars_and_all_ones(ctx, curve, group):
    """Adversarial bucket skew: every scalar equal (one bucket per window gets everything)."""
    c = get_curve(curve)
    G = Group(OC[curve], group)
    n = 300
    g_xy, _ = to_abi_points(curve, group, [G.gen])
    ds = list(range(1, n + 1))
    xy, inf = ctx.fixed_base_mul(c, group, g_xy, codec.fr_canonical(ds, c))
    bases = ctx.upload_bases(c, group, xy, inf)
    try:
        for k in (1, 0x1234567, c.r - 1):
            out = bases.msm(codec.fr_canonical([k] * n, c))
            assert jac_limbs_to_affine_oracle(curve, group, out) == G.mul(G.gen, k * sum(ds) % c.r)
    finally:
        bases.free()


@pytest.mark.parametrize("curve,group,log_n", [("bn254", 1, 12), ("bn254", 1, 16), ("bn254", 2, 14),
                                               ("bls12_381", 1, 14), ("bls12_381", 2, 12), ("bn254", 1, 20)])
def test_msm_known_dlog(ctx, curve, group, log_n):
    """bases P_i = d_i*G (built on the device), expected = (sum d_i k_i mod r)*G."""
    c = get_curve(curve)
    G = Group(OC[curve], group)
    n = (1 << log_n) - 3
    rng = np.random.default_rng(log_n * 10 + group)
    d = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    d[:, 3] >>= np.uint64(4)            # < 2^251 < r
    # FULL-RANGE scalars: uniform in [0, r) (256 random bits reduced mod r), so the top window's high digits and the last
    # signed-digit carry are exercised at every size; every 13th scalar sits within 2^20 of r - 1
    kraw = np.frombuffer(rng.bytes(32 * n), dtype=np.uint64).reshape(n, 4)
    kint = [x % c.r for x in codec.limbs_to_ints(kraw)]
    for i in range(0, n, 13):
        kint[i] = c.r - 1 - (kint[i] & 0xFFFFF)
    k = codec.fr_canonical(kint, c).reshape(n, 4).copy()
    assert max(kint).bit_length() == c.r.bit_length()
    k[::7] = 0                          # zero scalars (ark skips them)
    k[1::11, 1:] = 0
    k[1::11, 0] = 1                     # ones (ark fast path)
    g_xy, _ = to_abi_points(curve, group, [G.gen])
    xy, inf = ctx.fixed_base_mul(c, group, g_xy, d)
    assert not inf.any()
    bases = ctx.upload_bases(c, group, xy, inf)
    try:
        out = bases.msm(k)
        dl, kl = codec.limbs_to_ints(d), codec.limbs_to_ints(k)
        e = sum(a * b for a, b in zip(dl, kl)) % c.r
        assert jac_limbs_to_affine_oracle(curve, group, out) == G.mul(G.gen, e)
        # linearity: msm(k) + msm(k') == msm(k + k')  (scalars kept < r)
        k2 = np.zeros_like(k)
        k2[:, 0] = rng.integers(0, 1 << 62, size=n, dtype=np.uint64)
        ks = codec.fr_canonical([(a + b) % c.r for a, b in zip(kl, codec.limbs_to_ints(k2))], c)
        s1 = ctx.fold(c, group, np.concatenate([out, bases.msm(k2)]))
        assert jac_limbs_to_affine_oracle(curve, group, s1) == jac_limbs_to_affine_oracle(curve, group, bases.msm(ks))
    finally:
        bases.free()


def test_vartime_multiscalar_mul_montgomery_scalars(ctx):
    """zkp_curve::Curve::vartime_multiscalar_mul (curve/src/lib.rs:38-45): Fr scalars in Montgomery form."""
    c = get_curve("bn254")
    G = Group(OC["bn254"], 1)
    pts = random_points("bn254", 1, 40, seed=5)
    rnd = random.Random(9)
    ks = [rnd.randrange(c.r) for _ in range(40)]
    xy, inf = to_abi_points("bn254", 1, pts)
    bases = ctx.upload_bases(c, 1, xy, inf)
    try:
        out = bases.vartime_multiscalar_mul(codec.fr_to_mont(ks, c))
        assert jac_limbs_to_affine_oracle("bn254", 1, out) == G.msm_naive(pts, ks)
    finally:
        bases.free()


@pytest.mark.parametrize("curve,group", CFG)
def test_msm_repeatability_stress(ctx, curve, group):
    """Race detector: skewed inputs (long buckets -> split tasks, wave-level combines, LDS-atomic sort scatter) run 25
    times must give the identical group element every time and match the oracle once."""
    c = get_curve(curve)
    G = Group(OC[curve], group)
    rng = np.random.default_rng(77 + group)
    n = 3000
    d = rng.integers(1, 1 << 62, size=(n, 4), dtype=np.uint64)
    d[:, 3] >>= np.uint64(4)
    g_xy, _ = to_abi_points(curve, group, [G.gen])
    xy, inf = ctx.fixed_base_mul(c, group, g_xy, d)
    k = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    k[:, 3] >>= np.uint64(4)
    k[: n // 2] = k[0]                   # half the scalars identical -> 1500-entry buckets in every window
    k[n // 2: n // 2 + 700, 1:] = 0
    k[n // 2: n // 2 + 700, 0] = 1       # 700 ones -> one 700-entry bucket
    bases = ctx.upload_bases(c, group, xy, inf)
    try:
        first = jac_to_affine(ctx, curve, group, bases.msm(k))
        e = sum(a * b for a, b in zip(codec.limbs_to_ints(d), codec.limbs_to_ints(k))) % c.r
        assert first == G.mul(G.gen, e)
        for _ in range(24):
            assert jac_to_affine(ctx, curve, group, bases.msm(k)) == first
    finally:
        bases.free()


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_msm_skewed_scalars_known_dlog(ctx, curve):
    """SURVEY §8(d) skewed variant: half of the scalars in {0, 1} (boolean-heavy witnesses: one bucket of the first
    window receives a quarter of all entries), a block of identical scalars, and single-window scalars — against the
    known-discrete-log expectation at 2^18 points."""
    c = get_curve(curve)
    G = Group(OC[curve], 1)
    n = (1 << 18) - 5
    rng = np.random.default_rng(77)
    d = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    d[:, 3] >>= np.uint64(4)
    g_xy, _ = to_abi_points(curve, 1, [G.gen])
    xy, inf = ctx.fixed_base_mul(c, 1, g_xy, d)
    bases = ctx.upload_bases(c, 1, xy, inf)
    dl = codec.limbs_to_ints(d)
    try:
        k = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
        k[:, 3] >>= np.uint64(4)
        m = rng.random(n) < 0.5
        k[m] = 0
        k[m, 0] = rng.integers(0, 2, size=int(m.sum()), dtype=np.uint64)
        k[1000:60000] = k[999]                                  # a long run of identical scalars
        k[70000:90000, 1:] = 0
        k[70000:90000, 0] &= np.uint64(0xFF)                    # single-window scalars
        e = sum(a * b for a, b in zip(dl, codec.limbs_to_ints(k))) % c.r
        assert jac_limbs_to_affine_oracle(curve, 1, bases.msm(k)) == G.mul(G.gen, e)
    finally:
        bases.free()


@pytest.mark.parametrize("curve,group", [("bn254", 1), ("bn254", 2), ("bls12_381", 1), ("bls12_381", 2)])
def test_msm_random_lengths_and_offsets(ctx, curve, group):
    """Ragged inputs: random lengths (0 .. a few thousand, not powers of two), random offsets into the resident bases,
    lengths that overrun the bases (ark's min(len) truncation) — against the known-discrete-log expectation."""
    c = get_curve(curve)
    G = Group(OC[curve], group)
    nb = 3001
    rng = np.random.default_rng(1000 + group)
    d = rng.integers(0, 1 << 63, size=(nb, 4), dtype=np.uint64)
    d[:, 3] >>= np.uint64(4)
    g_xy, _ = to_abi_points(curve, group, [G.gen])
    xy, inf = ctx.fixed_base_mul(c, group, g_xy, d)
    bases = ctx.upload_bases(c, group, xy, inf)
    dl = codec.limbs_to_ints(d)
    try:
        for n, off in ((0, 0), (1, 0), (1, 3000), (2, 17), (63, 1), (64, 100), (65, 2936), (257, 5), (1000, 2001),
                       (3001, 0), (3500, 0), (500, 2900), (7, 3001)):
            k = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
            if n:
                k[:, 3] >>= np.uint64(4)
                k[rng.random(n) < 0.1] = 0
            m = max(0, min(n, nb - off))                      # terms that exist on both sides
            e = sum(a * b for a, b in zip(dl[off:off + m], codec.limbs_to_ints(k[:m]))) % c.r if m else 0
            got = jac_limbs_to_affine_oracle(curve, group, bases.msm(k, offset=off))
            assert got == (G.mul(G.gen, e) if e else None), (n, off)
    finally:
        bases.free()


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_vartime_multiscalar_mul_g2_and_mont_dev_vs_oracle(ctx, curve):
    """zkp_vartime_multiscalar_mul_g2 and zkp_msm_g2_mont_dev / zkp_msm_g1_mont_dev (device-resident Montgomery scalars,
    non-zero offset) against the oracle's naive MSM on oracle-generated bases, incl. an identity base and n > len."""
    c = get_curve(curve)
    rnd = random.Random(31)
    for group, n in ((2, 45), (1, 70)):
        G = Group(OC[curve], group)
        pts = random_points(curve, group, n, seed=40 + group)
        pts[2] = None
        ks = [rnd.randrange(c.r) for _ in range(n + 5)]
        ks[0], ks[1] = 0, c.r - 1
        xy, inf = to_abi_points(curve, group, pts)
        bases = ctx.upload_bases(c, group, xy, inf)
        try:
            km = codec.fr_to_mont(ks, c).reshape(-1, 4)
            out = bases.vartime_multiscalar_mul(km)                   # n + 5 scalars: truncated to the n bases
            assert jac_limbs_to_affine_oracle(curve, group, out) == G.msm_naive(pts, ks[:n])
            kd = ctx.to_device(km)
            for off, m in ((0, n), (7, n - 7), (n - 1, 1), (3, 20)):
                got = bases.msm_mont_dev(kd, m, offset=off)
                assert jac_limbs_to_affine_oracle(curve, group, got) == G.msm_naive(pts[off:off + m], ks[:m]), (group, off, m)
            ctx.dev_free(kd)
        finally:
            bases.free()


@pytest.mark.parametrize("curve,group,log_n", [("bn254", 1, 16), ("bn254", 2, 13)])
def test_msm_host_generated_bases_vs_cpu_oracle(ctx, curve, group, log_n):
    """Bases that never touched the device: built on the host by the C++ oracle's own fixed-base multiplication (and
    spot-checked against the Python big-int oracle), uniformly random canonical scalars; the device MSM must equal the
    oracle's window-parallel Pippenger (ark's algorithm) after normalisation."""
    from oracle import cpu_oracle
    c = get_curve(curve)
    G = Group(OC[curve], group)
    n = (1 << log_n) + 11
    rng = np.random.default_rng(2 * log_n + group)
    d = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    d[:, 3] >>= np.uint64(4)
    d[5] = 0                                                           # -> an identity base
    g_xy, _ = to_abi_points(curve, group, [G.gen])
    xy, inf = cpu_oracle.fixed_base_mul(c.cid, group, g_xy, d)
    assert inf[5] == 1 and inf.sum() == 1
    dl = codec.limbs_to_ints(d)
    for i in (0, 1, n - 1):
        want = to_abi_points(curve, group, [G.mul(G.gen, dl[i])])[0][0]
        assert np.array_equal(xy[i], want)
    k = np.frombuffer(rng.bytes(32 * n), dtype=np.uint64).reshape(-1, 4).copy()
    k[:, 3] &= np.uint64((1 << (c.r.bit_length() - 193)) - 1)          # < r
    bases = ctx.upload_bases(c, group, xy, inf)
    try:
        got = jac_limbs_to_affine_oracle(curve, group, bases.msm(k))
        want = jac_limbs_to_affine_oracle(curve, group, cpu_oracle.msm(c.cid, group, xy, inf, k, threads=8))
        assert got == want
        e = sum(a * b for a, b in zip(dl, codec.limbs_to_ints(k))) % c.r
        assert got == G.mul(G.gen, e)
    finally:
        bases.free()


@pytest.mark.parametrize("curve,group", CFG)
def test_msm_var_true_variable_base_small(ctx, curve, group):
    """zkp_msm_g*_var — `VariableBaseMSM::multi_scalar_mul(bases, scalars)` / `Curve::vartime_multiscalar_mul` with FRESH
    host bases (curve/src/lib.rs:38-45): no window tables, W separate bucket sets, doubling tail.  Edge cases of
    test_msm_small_edge_cases (c = 8 path), canonical and Montgomery scalars, against the oracle's naive MSM."""
    c = get_curve(curve)
    G = Group(OC[curve], group)
    rnd = random.Random(17)
    pts = random_points(curve, group, 24, seed=21 + group)
    pts[3] = None
    pts[5] = pts[4]
    pts[7] = G.neg(pts[6])
    ks = [rnd.randrange(c.r) for _ in range(24)]
    ks[0], ks[1], ks[2], ks[8], ks[9] = 0, 1, c.r - 1, 2, 1 << 253
    ks[5] = ks[4]
    ks[7] = ks[6]
    xy, inf = to_abi_points(curve, group, pts)
    for n in (24, 17, 1, 0):
        exp = G.msm_naive(pts[:n], ks[:n])
        out = ctx.msm_var(c, group, xy[:n], inf[:n], codec.fr_canonical(ks[:n], c).reshape(-1, 4))
        assert jac_limbs_to_affine_oracle(curve, group, out) == exp, n
        out = ctx.msm_var(c, group, xy[:n], inf[:n], codec.fr_to_mont(ks[:n], c).reshape(-1, 4), montgomery=True)
        assert jac_limbs_to_affine_oracle(curve, group, out) == exp, n
    # no identity flags at all (NULL), more scalars than points (min(len) truncation)
    out = ctx.msm_var(c, group, xy[:3], None, codec.fr_canonical(ks[:9], c).reshape(-1, 4))
    assert jac_limbs_to_affine_oracle(curve, group, out) == G.msm_naive(pts[:3], ks[:3])


@pytest.mark.parametrize("curve,group,log_n", [("bn254", 1, 13), ("bn254", 1, 17), ("bn254", 2, 13), ("bls12_381", 1, 14),
                                               ("bls12_381", 2, 12)])
def test_msm_var_equals_resident_msm_and_known_dlog(ctx, curve, group, log_n):
    """The variable-base path (c = 16, 16 bucket sets) against the known-discrete-log expectation and against the
    resident-table MSM on the same inputs, incl. zero / one scalars, an identity base and skewed (repeated) scalars."""
    c = get_curve(curve)
    G = Group(OC[curve], group)
    n = (1 << log_n) - 7
    rng = np.random.default_rng(log_n * 7 + group)
    d = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    d[:, 3] >>= np.uint64(4)
    d[9] = 0                                   # identity base
    k = np.frombuffer(rng.bytes(32 * n), dtype=np.uint64).reshape(-1, 4).copy()
    k[:, 3] &= np.uint64((1 << (c.r.bit_length() - 193)) - 1)
    k[::7] = 0
    k[1::11, 1:] = 0
    k[1::11, 0] = 1
    k[100:400] = k[99]                          # a 300-entry bucket in every window
    g_xy, _ = to_abi_points(curve, group, [G.gen])
    xy, inf = ctx.fixed_base_mul(c, group, g_xy, d)
    assert inf[9] == 1
    got = jac_limbs_to_affine_oracle(curve, group, ctx.msm_var(c, group, xy, inf, k))
    e = sum(a * b for a, b in zip(codec.limbs_to_ints(d), codec.limbs_to_ints(k))) % c.r
    assert got == G.mul(G.gen, e)
    bases = ctx.upload_bases(c, group, xy, inf)
    try:
        assert jac_limbs_to_affine_oracle(curve, group, bases.msm(k)) == got
    finally:
        bases.free()


@pytest.mark.parametrize("curve,group", CFG)
def test_msm_reduction_pyramid_exceptional_cases(ctx, curve, group):
    """Equal / opposite BUCKET sums meeting in the reduction pyramid (adjacent buckets holding Q and Q, or Q and -Q): the
    pairwise level must double / cancel exactly (bucket_dev.hpp hands P = +-Q to the exact formulas), in the resident-table
    path and in the variable-base path; plus sums that collide only at a higher level of the pyramid."""
    c = get_curve(curve)
    G = Group(OC[curve], group)
    Q, R = random_points(curve, group, 2, seed=77 + group)
    cases = [([Q, Q], [1, 2]),                       # buckets 0 and 1 both hold Q: level-0 pair is a doubling
             ([Q, G.neg(Q)], [1, 2]),                # Q and -Q: the pair cancels to the identity
             ([Q, R, Q, R], [1, 2, 3, 4]),           # (Q + R) meets (Q + R) one level up
             ([Q, R, G.neg(Q), G.neg(R)], [1, 2, 3, 4]),
             ([Q, Q, Q, Q, Q, Q, Q, Q], [1, 2, 3, 4, 5, 6, 7, 8])]   # every bucket holds the same point
    for pts, ks in cases:
        exp = G.msm_naive(pts, ks)
        xy, inf = to_abi_points(curve, group, pts)
        sc = codec.fr_canonical(ks, c).reshape(-1, 4)
        bases = ctx.upload_bases(c, group, xy, inf)
        try:
            assert jac_limbs_to_affine_oracle(curve, group, bases.msm(sc)) == exp, ks
        finally:
            bases.free()
        assert jac_limbs_to_affine_oracle(curve, group, ctx.msm_var(c, group, xy, inf, sc)) == exp, ks


@pytest.mark.parametrize("curve,group,log_n,distinct", [("bn254", 2, 18, 256), ("bn254", 1, 18, 256), ("bn254", 2, 19, 8192),
                                                         ("bls12_381", 2, 16, 64), ("bn254", 2, 20, 2048), ("bn254", 2, 22, 4096), ("bn254", 1, 22, 4096),
                                                         ("bn254", 2, 23, 2500), ("bn254", 1, 23, 2500)])   # > 32 tasks per bucket, > 2048 such buckets: every wave of combine_kernel walks two
def test_msm_many_long_buckets(ctx, curve, group, log_n, distinct):
    """Hundreds / thousands of buckets that are each split into many tasks (what the lowest buckets of a 2^24-point MSM look
    like: the short top window piles ~1500 entries onto each of them): scalars drawn from `distinct` small values, so
    `distinct` buckets of window 0 receive n / distinct entries each — the wave-per-bucket combine path on many waves at once.
    Known-discrete-log expectation."""
    c = get_curve(curve)
    G = Group(OC[curve], group)
    n = 1 << log_n
    rng = np.random.default_rng(log_n * 31 + group)
    d = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    d[:, 3] >>= np.uint64(4)
    g_xy, _ = to_abi_points(curve, group, [G.gen])
    xy, inf = ctx.fixed_base_mul(c, group, g_xy, d)
    k = np.zeros((n, 4), dtype=np.uint64)
    k[:, 0] = rng.integers(1, distinct + 1, size=n, dtype=np.uint64)
    bases = ctx.upload_bases(c, group, xy, inf)
    try:
        got = jac_limbs_to_affine_oracle(curve, group, bases.msm(k))
        e = sum(a * int(b) for a, b in zip(codec.limbs_to_ints(d), k[:, 0])) % c.r
        assert got == G.mul(G.gen, e)
    finally:
        bases.free()


@pytest.mark.parametrize("chunk,n", [(40000, (1 << 18) - 3), (1 << 16, (1 << 18) + 777), (100000, 300001)])
def test_chunked_msm_equals_unchunked_and_known_dlog(ctx, chunk, n):
    """Round 4: a large stand-alone G1 MSM is split by index into chunks that share one bucket array (bucket chaining) on two
    workspaces, so that the sort of chunk k + 1 runs under the accumulation of chunk k (msm.hip msm_run; zkp_ctx_config.msm_chunk_points:
    the chunked and the never-chunking context below share ONE resident table, zkp_bases_share).  Odd and even
    chunk counts, a ragged last chunk, zero / one / r - 1 scalars and identity bases: the result equals the unchunked MSM and the
    known-discrete-log expectation, call after call (the workspaces are reused)."""
    curve = "bn254"
    c = get_curve(curve)
    G = Group(OC[curve], 1)
    rng = np.random.default_rng(2026)
    d = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    d[:, 3] >>= np.uint64(4)
    d[5] = 0                                                     # an identity base
    g_xy, _ = to_abi_points(curve, 1, [G.gen])
    xy, inf = ctx.fixed_base_mul(c, 1, g_xy, d)
    assert inf[5] == 1
    from ckb_zkp_amd.api import Context
    plain_ctx = Context(ctx.device, dict(msm_chunk_points=-1))    # never chunks
    chunk_ctx = Context(ctx.device, dict(msm_chunk_points=chunk)) # chunks MSMs of >= 2 * chunk points
    assert plain_ctx.config()["msm_chunk_points"] == -1 and chunk_ctx.config()["msm_chunk_points"] == chunk
    bases0 = ctx.upload_bases(c, 1, xy, inf)
    plain_b, bases = bases0.share_with(plain_ctx), bases0.share_with(chunk_ctx)
    k = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    k[:, 3] >>= np.uint64(4)
    k[7] = 0
    k[8] = 0
    k[8, 0] = 1
    k[9] = codec.fr_canonical([c.r - 1], c)[0]
    k[n // 2: n // 2 + 5000] = k[n // 2]                          # a run of identical scalars across a chunk boundary region
    try:
        e = sum(a * b for a, b in zip(codec.limbs_to_ints(d), codec.limbs_to_ints(k))) % c.r
        want = G.mul(G.gen, e)
        plain = jac_limbs_to_affine_oracle(curve, 1, plain_b.msm(k))
        assert plain == want
        for _ in range(3):
            assert jac_limbs_to_affine_oracle(curve, 1, bases.msm(k)) == want
        # Montgomery-scalar batch entry point (PC::commit): two chunked MSMs in flight on partner workspaces + two plain ones
        km = codec.fr_to_mont(codec.limbs_to_ints(k), c).reshape(-1, 4)
        kd = ctx.to_device(km)
        outs = bases.msm_mont_batch_dev([(kd, n, 0), (kd, n, 0), (kd, 1000, 0), (kd + 32 * 10, n - 10, 10)])
        assert jac_limbs_to_affine_oracle(curve, 1, outs[0]) == want and jac_limbs_to_affine_oracle(curve, 1, outs[1]) == want
        ref = plain_b.msm_mont_batch_dev([(kd, 1000, 0), (kd + 32 * 10, n - 10, 10)])
        assert np.array_equal(jac_limbs_to_affine_oracle(curve, 1, outs[2]), jac_limbs_to_affine_oracle(curve, 1, ref[0]))
        assert jac_limbs_to_affine_oracle(curve, 1, outs[3]) == jac_limbs_to_affine_oracle(curve, 1, ref[1])
        ctx.dev_free(kd)
    finally:
        plain_b.free()
        bases.free()
        bases0.free()
        plain_ctx.close()
        chunk_ctx.close()
