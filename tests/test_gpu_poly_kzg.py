"""GPU: Marlin-side primitives (BASELINE.json configs[3] building blocks): Fr vector ops, batch inversion,
polynomial evaluation, division by (X - z), and KZG10 commit / open (marlin/src/pc/kzg10.rs:100-156) through the C ABI,
bit-exact against the big-int oracle; openings are additionally checked with the reference's own `KZG10::check`
pairing equation (kzg10.rs:158-173)."""
import random

import numpy as np
import pytest

from ckb_zkp_amd import api, codec, kzg10
from ckb_zkp_amd.params import get_curve
from oracle.pyref import kzg10 as okzg
from tests.util import OC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 1000, 8192 + 17, 300000])
def test_poly_eval_and_div_linear(ctx, curve, n):
    c = get_curve(curve)
    rnd = random.Random(n)
    p = [rnd.randrange(c.r) for _ in range(n)]
    if n > 40:
        p[5] = 0
        p[-1] = 1
    z = rnd.randrange(c.r)
    pm = codec.fr_to_mont(p, c).reshape(-1, 4)
    zm = codec.fr_to_mont([z], c)[0]
    dp = ctx.to_device(pm)
    dq = ctx.dev_alloc(max(n - 1, 1) * 32)
    try:
        assert codec.fr_from_mont(ctx.poly_evaluate(c, dp, n, zm).reshape(1, 4), c)[0] == okzg.evaluate(p, z, c.r)
        ev = ctx.poly_div_linear(c, dp, n, zm, dq)
        assert codec.fr_from_mont(ev.reshape(1, 4), c)[0] == okzg.evaluate(p, z, c.r)
        if n > 1:
            q = np.zeros((n - 1, 4), dtype=np.uint64)
            ctx.d2h(q, dq)
            assert codec.fr_from_mont(q, c) == okzg.divide_by_linear(p, z, c.r)
    finally:
        ctx.dev_free(dp)
        ctx.dev_free(dq)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_vec_ops_and_batch_inverse(ctx, curve):
    c = get_curve(curve)
    rnd = random.Random(3)
    n = 5000
    a = [rnd.randrange(c.r) for _ in range(n)]
    b = [rnd.randrange(c.r) for _ in range(n)]
    a[7] = 0
    k = rnd.randrange(c.r)
    da, db = ctx.to_device(codec.fr_to_mont(a, c)), ctx.to_device(codec.fr_to_mont(b, c))
    do = ctx.dev_alloc(n * 32)
    km = codec.fr_to_mont([k], c)[0]
    out = np.zeros((n, 4), dtype=np.uint64)
    try:
        exp = {api.VEC_MUL: [x * y % c.r for x, y in zip(a, b)], api.VEC_ADD: [(x + y) % c.r for x, y in zip(a, b)],
               api.VEC_SUB: [(x - y) % c.r for x, y in zip(a, b)], api.VEC_SCALE: [k * x % c.r for x in a],
               api.VEC_AXPY: [(x + k * y) % c.r for x, y in zip(a, b)]}
        for op, e in exp.items():
            ctx.fr_vec_op(c, op, da, db, do, n, km)
            ctx.d2h(out, do)
            assert codec.fr_from_mont(out, c) == e, op
        ctx.fr_batch_inverse(c, da, n)
        ctx.d2h(out, da)
        assert codec.fr_from_mont(out, c) == [pow(x, -1, c.r) if x else 0 for x in a]
    finally:
        for d in (da, db, do):
            ctx.dev_free(d)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_kzg10_commit_open_check(ctx, curve):
    """Mirror of marlin/src/pc/kzg10.rs tests (commit -> open -> check), hiding and non-hiding, with leading zeros."""
    c = get_curve(curve)
    rnd = random.Random(11)
    deg, beta = 50, 0x123456789ABCDEF0123
    pp = okzg.setup(OC[curve], deg, beta)
    ck = kzg10.setup(ctx, curve, deg, beta)
    assert codec.g1_from_mont(*ck.host_g, c) == pp["powers_of_g"]
    try:
        for lead_zeros, hiding in ((0, False), (3, False), (0, True), (2, True)):
            p = [0] * lead_zeros + [rnd.randrange(c.r) for _ in range(deg + 1 - lead_zeros)]
            blind = [rnd.randrange(c.r) for _ in range(3)] if hiding else None
            z = rnd.randrange(c.r)
            pm = codec.fr_to_mont(p, c).reshape(-1, 4)
            bm = codec.fr_to_mont(blind, c).reshape(-1, 4) if hiding else None
            comm = kzg10.commit(ctx, ck, pm, bm)
            assert comm == okzg.commit(pp, p, blind)
            w, rand_v = kzg10.open(ctx, ck, pm, z, bm)
            assert (w, rand_v) == okzg.open_(pp, p, z, blind)
            assert okzg.check(pp, comm, z, okzg.evaluate(p, z, c.r), w, rand_v)            # reference's acceptance test
            assert not okzg.check(pp, comm, z, (okzg.evaluate(p, z, c.r) + 1) % c.r, w, rand_v)
        with pytest.raises(kzg10.KzgError):
            kzg10.commit(ctx, ck, codec.fr_to_mont([5], c).reshape(-1, 4))                   # DegreeIsZero
    finally:
        ck.powers_of_g.free()
        ck.powers_of_gamma_g.free()


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_batched_commit_msms_equal_individual_ones(ctx, curve):
    """zkp_msm_g1_mont_batch_dev (PC::commit over a list: concurrent MSMs on the context's streams) returns, job by job,
    the same group element as zkp_msm_g1_mont_dev — including offsets, truncation at the end of the powers and n = 0."""
    c = get_curve(curve)
    rnd = random.Random(17)
    ck = kzg10.setup(ctx, curve, 700, 0x5EED5EED)
    try:
        jobs, ptrs = [], []
        for n, off in ((701, 0), (300, 401), (2, 0), (0, 5), (650, 100), (1, 700), (257, 13), (64, 0)):
            coeffs = codec.fr_to_mont([rnd.randrange(c.r) for _ in range(max(n, 1))], c).reshape(-1, 4)
            d = ctx.to_device(coeffs)
            ptrs.append(d)
            jobs.append((d, n, off))
        got = ck.powers_of_g.msm_mont_batch_dev(jobs)
        for k, (d, n, off) in enumerate(jobs):
            want = ck.powers_of_g.msm_mont_dev(d, n, offset=off)
            assert ctx.into_affine(c, 1, got[k])[0].tolist() == ctx.into_affine(c, 1, want)[0].tolist(), (k, n, off)
            assert int(ctx.into_affine(c, 1, got[k])[1]) == int(ctx.into_affine(c, 1, want)[1])
        for d in ptrs:
            ctx.dev_free(d)
    finally:
        ck.powers_of_g.free()
        ck.powers_of_gamma_g.free()


def test_shared_bases_between_contexts(ctx):
    """zkp_bases_share: a second context (own streams and scratch) computes the same MSM against the first context's
    resident window tables; freeing one handle leaves the other usable."""
    from ckb_zkp_amd.api import Context
    c = get_curve("bn254")
    rnd = random.Random(23)
    ck = kzg10.setup(ctx, "bn254", 300, 0xC0DE)
    ctx2 = Context(0)
    try:
        shared = ck.powers_of_g.share_with(ctx2)
        coeffs = codec.fr_to_mont([rnd.randrange(c.r) for _ in range(301)], c).reshape(-1, 4)
        d1, d2 = ctx.to_device(coeffs), ctx2.to_device(coeffs)
        a = ck.powers_of_g.msm_mont_dev(d1, 301)
        b = shared.msm_mont_dev(d2, 301)
        assert ctx.into_affine(c, 1, a)[0].tolist() == ctx2.into_affine(c, 1, b)[0].tolist()
        ck.powers_of_g.free()
        b2 = shared.msm_mont_dev(d2, 301)                     # the tables outlive the first handle
        assert ctx2.into_affine(c, 1, b2)[0].tolist() == ctx2.into_affine(c, 1, b)[0].tolist()
        shared.free()
        ctx.dev_free(d1)
        ctx2.dev_free(d2)
    finally:
        ck.powers_of_gamma_g.free()
        ctx2.close()
