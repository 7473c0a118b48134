"""Seeded differential fuzz of the two primitives the path is made of: random shapes, random identity
patterns, adversarial scalar classes — HIP through the C ABI against oracle/cpu (the C++ restatement of ark-ec's window-parallel
Pippenger and ark-poly's radix-2 transforms), both as group elements / field elements, bit for bit.  Complements the hand-picked
edge sets of test_gpu_msm.py / test_gpu_ntt.py; 480 MSM cases (+ 120 through the variable-base entry point) and 480 transforms, ≈ 25 s on the GPU box."""
import numpy as np
import pytest

from ckb_zkp_amd import codec
from ckb_zkp_amd.api import NTT_COSET_FFT, NTT_COSET_IFFT, NTT_FFT, NTT_IFFT
from ckb_zkp_amd.params import get_curve
from oracle import cpu_oracle
from oracle.pyref.curves import Group
from tests.util import OC, jac_limbs_to_affine_oracle, to_abi_points

pytestmark = pytest.mark.gpu


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
    rng = np.random.default_rng(0xF00D + 17 * group + c.cid)
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
    rng = np.random.default_rng(0xBEEF + c.cid)
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
