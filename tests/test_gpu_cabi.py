"""GPU: the C ABI driven from a plain C99 program (tests/c/abi_driver.c) — NTT round trip, MSM through host and device
entry points, error statuses — with inputs from / results checked against the oracle."""
import random
import subprocess

import numpy as np
import pytest

from ckb_zkp_amd import codec
from ckb_zkp_amd.params import get_curve
from oracle.pyref import ntt as ontt
from oracle.pyref.curves import Group
from tests.util import OC, jac_limbs_to_affine_oracle, random_points, to_abi_points

pytestmark = pytest.mark.gpu


def test_c_program_ntt_and_msm_match_the_oracle(tmp_path):
    from tests import c_driver
    exe = c_driver.build()
    curve, log_n, npts = "bn254", 9, 33
    c = get_curve(curve)
    G = Group(OC[curve], 1)
    rnd = random.Random(5)
    vals = [rnd.randrange(c.r) for _ in range(1 << log_n)]
    pts = random_points(curve, 1, npts, seed=8)
    pts[4] = None
    ks = [rnd.randrange(c.r) for _ in range(npts)]
    xy, inf = to_abi_points(curve, 1, pts)
    infw = np.zeros((npts + 7) // 8 * 8, dtype=np.uint8)
    infw[:npts] = inf
    blob = b"".join([np.array([c.cid, log_n, npts, npts], dtype=np.uint64).tobytes(),
                     codec.fr_to_mont(vals, c).tobytes(), np.ascontiguousarray(xy, dtype=np.uint64).tobytes(),
                     infw.tobytes(), codec.fr_canonical(ks, c).tobytes()])
    (tmp_path / "in.bin").write_bytes(blob)
    r = subprocess.run([str(exe), "run", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
    out = np.frombuffer((tmp_path / "out.bin").read_bytes(), dtype=np.uint64)
    n = 1 << log_n
    a, b = out[:4 * n].reshape(n, 4), out[4 * n:8 * n].reshape(n, 4)
    rest = out[8 * n:]
    assert codec.fr_from_mont(a, c) == ontt.Domain(OC[curve], n).coset_fft(vals)
    assert codec.fr_from_mont(b, c) == vals
    want = G.msm_naive(pts, ks)
    assert codec.g1_from_mont(rest[:8].reshape(1, 8), [int(rest[8])], c)[0] == want
    assert jac_limbs_to_affine_oracle(curve, 1, rest[9:21]) == want
