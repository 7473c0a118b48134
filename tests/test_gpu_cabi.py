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


# ------------------------------------------------------------------------------------------------ the prover seams, from C
def _sec(a) -> bytes:
    """one section of the driver's input: [u64 byte count][payload zero-padded to 8 bytes]"""
    b = a if isinstance(a, (bytes, bytearray)) else np.ascontiguousarray(a).tobytes()
    return len(b).to_bytes(8, "little") + bytes(b) + bytes(-len(b) % 8)


def _run_driver(mode, blob, tmp_path):
    from tests import c_driver
    exe = c_driver.build()
    (tmp_path / "in.bin").write_bytes(blob)
    r = subprocess.run([str(exe), mode, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
    return (tmp_path / "out.bin").read_bytes()


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_c_program_groth16_seam_reproduces_the_golden_proofs(tmp_path, curve):
    """(Since round 6 the program also creates a second context with zkp_ctx_create_ex — two lanes, coefficient-form key, device-side
    into_affine — reads the configuration back and requires the same proof and the same batch: compared in C.)
    zkp_groth16_pk_upload -> zkp_groth16_witness_map -> zkp_groth16_prove -> zkp_groth16_prove_batch from a plain C99
    program (what the Rust `create_proof` seam of rust/patches binds, groth16/src/prover.rs:124) on the golden Mini / MiMC
    instances: h and the proof of the golden (r, s) equal tests/golden/golden.json, the other proofs of the batch equal the
    oracle prover's.  The key arrays are built from the ORACLE's key on the host: no ctypes call touches the library here."""
    from ckb_zkp_amd.r1cs import ConstraintSystem, R1csInstance
    from oracle.pyref import groth16 as og
    from tests.golden_util import GOLDEN, I, TOXIC, abi_params_from_oracle, golden_circuits, unpt
    c = get_curve(curve)
    f = c.fq_limbs
    for e in GOLDEN["curves"][curve]["groth16"]:
        ocirc, ocirc_setup, pcirc, _ = golden_circuits(curve, e)
        opk = og.generate_parameters(OC[curve], ocirc_setup, **TOXIC, g1_k=e["g1_k"], g2_k=e["g2_k"])
        cs = ConstraintSystem(curve, True)
        pcirc.generate_constraints(cs)
        inst = R1csInstance.from_cs(cs)
        P = abi_params_from_oracle(curve, opk, inst.num_inputs, inst.num_aux, inst.num_constraints())
        rnd = random.Random(77)
        rs = [(I(e["r"]), I(e["s"]))] + [(rnd.randrange(c.r), rnd.randrange(c.r)) for _ in range(3)] + [(0, 0)]
        blob = _sec(np.array([c.cid, inst.num_inputs, inst.num_aux, inst.num_constraints(), len(rs)], dtype=np.uint64))
        for which in "abc":
            rp, col, cf = inst.csr(which)
            blob += _sec(rp.astype(np.uint32)) + _sec(col.astype(np.uint32)) + _sec(cf.astype(np.uint64))
        for name in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2"):
            blob += _sec(getattr(P, name).astype(np.uint64))
        for name in ("a_query", "b_g1_query", "b_g2_query", "h_query", "l_query"):
            xy, inf = getattr(P, name)
            blob += _sec(xy.astype(np.uint64)) + _sec(np.asarray(inf, dtype=np.uint8))
        blob += _sec(codec.fr_to_mont(inst.z, c)) + _sec(codec.fr_to_mont([r for r, _ in rs], c)) + \
            _sec(codec.fr_to_mont([s for _, s in rs], c))
        out = np.frombuffer(_run_driver("groth16", blob, tmp_path), dtype=np.uint64)
        N = int(out[0])
        assert N == len(e["h"])
        assert codec.fr_from_mont(out[1:1 + 4 * N].reshape(N, 4), c) == [I(v) for v in e["h"]]
        rest = out[1 + 4 * N:].reshape(-1, 8 * f + 3)
        assert rest.shape[0] == len(rs) + 3

        def dec(row):
            a = codec.g1_from_mont(row[:2 * f].reshape(1, -1), [int(row[8 * f])], c)[0]
            b = codec.g2_from_mont(row[2 * f:6 * f].reshape(1, -1), [int(row[8 * f + 1])], c)[0]
            cc = codec.g1_from_mont(row[6 * f:8 * f].reshape(1, -1), [int(row[8 * f + 2])], c)[0]
            return a, b, cc
        golden = (unpt(e["a"], 1), unpt(e["b"], 2), unpt(e["c"], 1))
        assert dec(rest[0]) == golden                              # zkp_groth16_prove
        assert dec(rest[1]) == golden                              # zkp_groth16_prove_batch, proof 0
        assert dec(rest[-2]) == golden                             # zkp_groth16_prove again after the batch
        assert dec(rest[-1]) == golden                             # zkp_groth16_prove_multi, 3 ranks on device 0 (also compared in C)
        for k in range(1, len(rs)):
            op, _ = og.create_proof(opk, ocirc, *rs[k])
            assert dec(rest[1 + k]) == (op.a, op.b, op.c), k      # incl. (r, s) = (0, 0): create_proof_no_zk


def test_c_program_marlin_seam_matches_the_oracle_prover(tmp_path):
    """zkp_bases_upload_g1 (SRS) -> zkp_marlin_index_upload -> zkp_marlin_index_commit -> zkp_marlin_prove from the C99 program
    (the seam of marlin::create_random_proof, marlin/src/lib.rs:97-181) on the 3-sample MiMC chain: index commitments, the
    transcript-derived challenges, commitments, 21 evaluations and both opening proofs equal the oracle's create_random_proof,
    and the oracle's verifier accepts.  SRS, matrices and the key bytes are produced by the oracle / host code only."""
    import ctypes as C
    from ckb_zkp_amd import _lib
    from ckb_zkp_amd import marlin as pm
    from ckb_zkp_amd.circuits import mimc_chain_instance
    from oracle.pyref import fs_rng as ofs
    from oracle.pyref import groth16 as og
    from oracle.pyref import kzg10 as okzg
    from oracle.pyref import marlin as om
    curve = "bn254"
    c = get_curve(curve)
    inst = mimc_chain_instance(curve, 3, seed=91)
    ocirc = og.MimcChain(OC[curve], inst.constants, inst.preimages)
    oidx = om.index(OC[curve], ocirc)
    beta_srs = 0x13579BDF2468ACE
    pp = okzg.setup(OC[curve], oidx["max_degree"], beta_srs)
    ic = om.index_commitments(oidx, pp)
    ivk = om.index_verifier_key(oidx, pp, ic)
    n, pad_aux, mats = pm.prepare_matrices(inst)
    rnd = random.Random(57)
    hs = oidx["dh"].size
    R = dict(w=[rnd.randrange(c.r)], z_a=[rnd.randrange(c.r)], z_b=[rnd.randrange(c.r)],
             mask=[rnd.randrange(c.r) for _ in range(3 * hs)],
             blind={l: [rnd.randrange(c.r), rnd.randrange(c.r)] for l in ("w", "z_a", "z_b", "g_1")},
             blind_shifted={"g_1": [rnd.randrange(c.r), rnd.randrange(c.r)]})
    x, w = inst.z[:inst.num_inputs], inst.z[inst.num_inputs:]
    g_xy, _ = to_abi_points(curve, 1, pp["powers_of_g"])
    gg_xy, _ = to_abi_points(curve, 1, pp["powers_of_gamma_g"])
    mont = lambda v: codec.fr_to_mont(list(v), c)
    blob = _sec(np.array([c.cid, inst.num_inputs, n, pad_aux, len(w), len(pp["powers_of_g"]), len(pp["powers_of_gamma_g"])],
                         dtype=np.uint64))
    for ptr, col, cf, _rows in mats:
        blob += _sec(np.asarray(ptr, dtype=np.uint32)) + _sec(np.asarray(col, dtype=np.uint32)) + _sec(np.asarray(cf, dtype=np.uint64))
    blob += _sec(g_xy.astype(np.uint64)) + _sec(gg_xy.astype(np.uint64)) + _sec(ofs.index_verifier_key_bytes(ivk, OC[curve]))
    blob += _sec(mont(x)) + _sec(mont(w)) + _sec(mont(R["w"])) + _sec(mont(R["z_a"])) + _sec(mont(R["z_b"])) + _sec(mont(R["mask"]))
    for l in ("w", "z_a", "z_b", "g_1"):
        blob += _sec(mont(R["blind"][l]))
    blob += _sec(mont(R["blind_shifted"]["g_1"]))
    raw = _run_driver("marlin", blob, tmp_path)
    info = np.frombuffer(raw[:48], dtype=np.uint64)
    assert [int(v) for v in info] == [oidx["dx"].size, hs, oidx["dk"].size, oidx["db"].size, oidx["max_degree"],
                                      oidx["num_non_zeros"]]
    sz = int.from_bytes(raw[48:56], "little")
    assert sz == C.sizeof(_lib.MarlinProof)                       # the C struct and the ctypes mirror agree on the layout
    pr = _lib.MarlinProof.from_buffer_copy(raw[56:56 + sz])
    tail = np.frombuffer(raw[56 + sz:], dtype=np.uint64)
    w2 = 2 * c.fq_limbs
    g1 = lambda words, inf: codec.g1_from_mont(np.array(words[:w2], dtype=np.uint64).reshape(1, w2), [int(inf)], c)[0]
    icomm = tail[:144].reshape(12, 12)
    assert {l: (g1(icomm[k], tail[144 + k]), None) for k, l in enumerate(om.INDEX_LABELS)} == ic
    o = om.create_random_proof(oidx, pp, ic, ocirc, R)
    chs = codec.fr_from_mont(np.array(pr.challenges, dtype=np.uint64).reshape(7, 4), c)
    assert dict(zip(("alpha", "eta_a", "eta_b", "eta_c", "beta", "gamma", "xi"), chs)) == o["challenges"]
    comm = np.array(pr.comm, dtype=np.uint64).reshape(9, 12)
    sh = np.array(pr.shifted, dtype=np.uint64).reshape(2, 12)
    comms = {}
    for i, l in enumerate(pm.LABELS_1 + pm.LABELS_2 + pm.LABELS_3):
        s = g1(sh[0], pr.shifted_inf[0]) if l == "g_1" else g1(sh[1], pr.shifted_inf[1]) if l == "g_2" else None
        comms[l] = (g1(comm[i], pr.comm_inf[i]), s)
    assert comms == o["commitments"]
    evals = codec.fr_from_mont(np.array(pr.evaluations, dtype=np.uint64).reshape(-1, 4), c)
    assert evals == o["evaluations"]
    ow = np.array(pr.opening_w, dtype=np.uint64).reshape(2, 12)
    rv = codec.fr_from_mont(np.array(pr.opening_rand_v, dtype=np.uint64).reshape(2, 4), c)
    proofs = [(g1(ow[k], pr.opening_w_inf[k]), rv[k] if pr.opening_has_rand[k] else None) for k in range(pr.num_opening_proofs)]
    assert proofs == o["opening_proofs"]
    assert om.verify_random_proof(oidx, pp, ic, dict(commitments=comms, evaluations=evals, opening_proofs=proofs), x[1:])
