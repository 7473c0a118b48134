"""Kernel-level oracle equality at BASELINE.json's sizes (SURVEY §8(d) "Kernel-level inputs": MSM n and NTT N up to 2^24,
"round-trip + oracle compare").  The full-size proofs cover these kernels in composition (fused chains, evaluation-form key);
here the stand-alone entry points are compared directly:

  zkp_ntt   all four ops == oracle/cpu's radix-2 transform on EVERY output, 2^20 / 2^22 / 2^24, both scalar fields (2^22 and up
            run the three-pass 9-radix-bit plan of ntt.hip)                                   r1cs_to_qap.rs:144-169
  zkp_msm   known discrete logs: bases d_i*G from the device's fixed-base kernel, expected (sum d_i k_i)*G by the oracle's
            inner product + the Python group law; BN254 G1 at 2^22 / 2^24, G2 and BLS12-381 G1 / G2 at 2^20; at 2^20 also ==
            oracle/cpu's Pippenger (ark's window rule)                                           prover.rs:187,190,220

ZKP_TEST_FULL=0 keeps one representative per kernel (NTT 2^22 BN254, MSM 2^22 BN254 G1) for a suite that must fit a time limit."""
import os

import numpy as np
import pytest

from ckb_zkp_amd import api, codec
from ckb_zkp_amd.params import get_curve
from oracle import cpu_oracle
from oracle.pyref.curves import Group
from tests.util import OC, jac_limbs_to_affine_oracle, to_abi_points

pytestmark = pytest.mark.gpu
from tests.util import TEST_FULL as FULL
OPS = (api.NTT_FFT, api.NTT_IFFT, api.NTT_COSET_FFT, api.NTT_COSET_IFFT)


def uniform_below_r(rng, n, c):
    """(n, 4) limbs of integers spread over [0, r): 192 random low bits, top limb uniform below r's top limb"""
    a = np.frombuffer(rng.bytes(32 * n), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] = rng.integers(0, c.r >> 192, size=n, dtype=np.uint64)
    return a


NTT_CASES = [("bn254", 20), ("bn254", 22), ("bn254", 24), ("bls12_381", 20), ("bls12_381", 22), ("bls12_381", 24)]


@pytest.mark.parametrize("curve,log_n", NTT_CASES if FULL else [("bn254", 22)])
def test_ntt_every_output_equals_cpu_oracle(ctx, curve, log_n):
    c = get_curve(curve)
    n = 1 << log_n
    x = uniform_below_r(np.random.default_rng(300 + log_n), n, c)          # valid Montgomery residues over the whole range
    x[0], x[1] = 0, codec.fr_to_mont([c.r - 1], c)[0]
    threads = cpu_oracle.hardware_threads()
    for op in OPS:
        got = ctx.ntt(c, x, op)
        exp = cpu_oracle.ntt(OC[curve].cid, x, op, threads=threads)
        assert np.array_equal(got, exp), (curve, log_n, op, int(np.flatnonzero((got != exp).any(axis=1))[0]))
        del got, exp


MSM_CASES = [("bn254", 1, 22), ("bn254", 1, 24), ("bn254", 2, 20), ("bls12_381", 1, 20), ("bls12_381", 2, 20)]


@pytest.mark.parametrize("curve,group,log_n", MSM_CASES if FULL else [("bn254", 1, 22)])
def test_msm_known_dlog_at_baseline_sizes(ctx, curve, group, log_n):
    c = get_curve(curve)
    oc = OC[curve]
    G = Group(oc, group)
    n = (1 << log_n) - 5
    rng = np.random.default_rng(900 + 10 * log_n + group)
    d = np.frombuffer(rng.bytes(32 * n), dtype=np.uint64).reshape(n, 4).copy()
    d[:, 3] >>= np.uint64(5)                                             # discrete logs of the bases: < 2^251 < r, nonzero
    d[:, 0] |= np.uint64(1)
    k = uniform_below_r(rng, n, c)                                       # full-range scalars: top window + last signed-digit carry
    top = np.frombuffer((c.r - 1).to_bytes(32, "little"), dtype="<u8")
    k[::13] = top
    k[::13, 0] -= rng.integers(0, 1 << 20, size=len(k[::13]), dtype=np.uint64)      # within 2^20 of r - 1 (low limb of r - 1 >= 2^28)
    k[::7] = 0                                                           # ark skips zeros
    k[1::11, 1:] = 0
    k[1::11, 0] = 1                                                      # ark's fast path for ones
    # expected exponent: sum d_i k_i mod r.  oracle fr_dot multiplies Montgomery-style (a b / R per term) and converts once
    # more (/ R): canonical inputs in -> sum / R^2 out.  Pinned on a prefix with Python integers.
    R2 = pow(1 << 256, 2, c.r)
    m = 999
    assert cpu_oracle.fr_dot(oc, d[:m], k[:m]) * R2 % c.r == \
        sum(a * b for a, b in zip(codec.limbs_to_ints(d[:m]), codec.limbs_to_ints(k[:m]))) % c.r
    e = cpu_oracle.fr_dot(oc, d, k) * R2 % c.r
    g_xy, _ = to_abi_points(curve, group, [G.gen])
    xy, inf = ctx.fixed_base_mul(c, group, g_xy, d)
    assert not inf.any()
    bases = ctx.upload_bases(c, group, xy, inf)
    try:
        out = bases.msm(k)
        assert jac_limbs_to_affine_oracle(curve, group, out) == G.mul(G.gen, e), (curve, group, log_n)
        if log_n <= 20:
            ref = cpu_oracle.msm(oc.cid, group, xy, inf, k, threads=cpu_oracle.hardware_threads())
            assert jac_limbs_to_affine_oracle(curve, group, ref) == jac_limbs_to_affine_oracle(curve, group, out)
    finally:
        bases.free()
