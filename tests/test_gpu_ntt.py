"""NTT parity: HIP Stockham NTT (through the C ABI) vs the big-int oracle, bit-exact on Montgomery limbs.
Mirrors the ark-poly calls of /root/reference/groth16/src/r1cs_to_qap.rs:144-169."""
import random

import numpy as np
import pytest

from ckb_zkp_amd import api, codec
from ckb_zkp_amd.params import get_curve
from oracle.pyref.ntt import Domain
from tests.util import OC

pytestmark = pytest.mark.gpu
OPS = {api.NTT_FFT: "fft", api.NTT_IFFT: "ifft", api.NTT_COSET_FFT: "coset_fft", api.NTT_COSET_IFFT: "coset_ifft"}


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 7, 8, 10, 11, 13])
def test_ntt_matches_oracle(ctx, curve, log_n):
    c = get_curve(curve)
    rnd = random.Random(1000 + log_n)
    n = 1 << log_n
    x = [rnd.randrange(c.r) for _ in range(n)]
    if n >= 4:
        x[0], x[1], x[2] = 0, 1, c.r - 1
    d = Domain(OC[curve], n)
    xm = codec.fr_to_mont(x, c)
    for op, name in OPS.items():
        got = codec.fr_from_mont(ctx.ntt(c, xm, op), c)
        assert got == getattr(d, name)(x), (curve, log_n, name)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
@pytest.mark.parametrize("log_n", [16, 20, 22])
def test_ntt_roundtrip_and_linearity_large(ctx, curve, log_n):
    """Size-independent properties at BASELINE sizes: ifft(fft(x)) == x, coset round trip, linearity,
    and out[0] == sum(x) (DC term) checked against a Python big-int sum."""
    c = get_curve(curve)
    n = 1 << log_n
    rng = np.random.default_rng(log_n)
    raw = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)      # < 2^254 < r: valid Montgomery residues
    raw[:, 3] &= np.uint64((1 << 60) - 1)
    y = ctx.ntt(c, raw, api.NTT_FFT)
    assert np.array_equal(ctx.ntt(c, y, api.NTT_IFFT), raw)
    yc = ctx.ntt(c, raw, api.NTT_COSET_FFT)
    assert np.array_equal(ctx.ntt(c, yc, api.NTT_COSET_IFFT), raw)
    assert not np.array_equal(y, yc)
    # DC term
    vals = codec.limbs_to_ints(raw)
    Ri = pow(1 << 256, -1, c.r)
    assert codec.fr_from_mont(y[:1], c)[0] == sum(vals) % c.r * Ri % c.r
    # spot-check one more output against the definition with a stride-subsampled input (x supported on a coset)
    k = 5
    wk = pow(pow(pow(c.fr_generator, (c.r - 1) >> c.two_adicity, c.r), 1 << (c.two_adicity - log_n), c.r), k, c.r)
    acc, p = 0, 1
    for v in vals:
        acc = (acc + v * p) % c.r
        p = p * wk % c.r
    assert codec.fr_from_mont(y[k:k + 1], c)[0] == acc * Ri % c.r


def test_ntt_domain_too_large(ctx):
    """EvaluationDomain::new -> None -> PolynomialDegreeTooLarge (r1cs_to_qap.rs:123-125)."""
    from ckb_zkp_amd._lib import ZkpError
    a = np.zeros((2, 4), dtype=np.uint64)
    with pytest.raises(ZkpError) as e:
        ctx.lib.zkp_ntt  # noqa: B018
        from ckb_zkp_amd import _lib
        _lib.check(ctx.lib.zkp_ntt(ctx.h, 0, a.ctypes.data, 29, 0), "zkp_ntt")
    assert e.value.status == -3
