"""CPU, world_size 2, gloo: the multi-GPU path of BASELINE.json configs[4] — bases sharded by index, partial MSMs,
all-gather of the partial points, local fold — with an oracle-backed engine standing in for the GPU
(`ckb_zkp_amd.distributed` is device-agnostic; the GPU engine is covered by tests/test_gpu_dist.py).
Checks: shard bounds tile the range; sharded MSM == unsharded oracle MSM on every rank; sharded Groth16 sums fold to
the same five points as the single-process oracle prover's, and the proof assembled from them is the golden proof."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleEngine:
    """Test stand-in for the device: oracle/cpu restatement (tests may use the oracle as the checker)."""

    class H:
        def __init__(self, curve, group, xy, inf):
            self.curve, self.group, self.xy, self.inf = curve, group, xy, inf

    def upload(self, curve, group, xy, inf):
        return OracleEngine.H(curve, group, xy, inf)

    def msm(self, h, scalars_mont):
        from ckb_zkp_amd import codec
        from oracle import cpu_oracle
        sc = codec.fr_canonical(codec.fr_from_mont(scalars_mont, h.curve), h.curve) if len(scalars_mont) else \
            np.zeros((0, 4), dtype=np.uint64)
        return cpu_oracle.msm(h.curve.cid, h.group, h.xy, h.inf, sc)

    def fold(self, curve, group, stacked):
        # sum of k Jacobian points == MSM of their affine forms with scalar 1
        from ckb_zkp_amd import codec
        from oracle import cpu_oracle
        from tests.util import jac_limbs_to_affine_oracle, to_abi_points
        w = 3 * curve.fq_limbs * group
        pts = [jac_limbs_to_affine_oracle(curve.name, group, stacked[i:i + w]) for i in range(0, len(stacked), w)]
        xy, inf = to_abi_points(curve.name, group, pts)
        return cpu_oracle.msm(curve.cid, group, xy, inf, codec.fr_canonical([1] * len(pts), curve))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ckb_zkp_amd import codec
        from ckb_zkp_amd.distributed import ShardedBases, ShardedGroth16Prover, all_gather_points
        from ckb_zkp_amd.params import get_curve
        from ckb_zkp_amd.r1cs import ConstraintSystem, R1csInstance
        from oracle import cpu_oracle
        from oracle.pyref import groth16 as og
        from tests.golden_util import GOLDEN, I, TOXIC, abi_params_from_oracle, golden_circuits, unpt
        from tests.util import OC, jac_limbs_to_affine_oracle, random_points, to_abi_points
        import random
        curve = "bn254"
        c = get_curve(curve)
        eng = OracleEngine()
        # --- sharded MSM, n not divisible by world, fewer scalars than bases
        for group, n in ((1, 37), (2, 9)):
            pts = random_points(curve, group, n, seed=3)
            pts[1] = None
            rnd = random.Random(4)
            ks = [rnd.randrange(c.r) for _ in range(n - 2)]
            xy, inf = to_abi_points(curve, group, pts)
            sb = ShardedBases(eng, c, group, xy, inf, rank, world)
            got = sb.msm(codec.fr_to_mont(ks, c).reshape(-1, 4))
            full = cpu_oracle.msm(c.cid, group, xy, inf, codec.fr_canonical(ks, c))
            assert jac_limbs_to_affine_oracle(curve, group, got) == jac_limbs_to_affine_oracle(curve, group, full)
        # --- sharded Groth16 on the golden MiMC instance
        e = GOLDEN["curves"][curve]["groth16"][1]
        ocirc, ocirc_setup, pcirc, _ = golden_circuits(curve, e)
        opk = og.generate_parameters(OC[curve], ocirc_setup, **TOXIC, g1_k=e["g1_k"], g2_k=e["g2_k"])
        cs = ConstraintSystem(curve, True)
        pcirc.generate_constraints(cs)
        inst = R1csInstance.from_cs(cs)
        params = abi_params_from_oracle(curve, opk, inst.num_inputs, inst.num_aux, inst.num_constraints())
        z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
        wm = lambda zz: cpu_oracle.witness_map(params, inst, zz)
        prover = ShardedGroth16Prover(eng, params, inst, rank, world, witness_mapper=wm)
        r_, s_ = I(e["r"]), I(e["s"])
        sums = prover.prove_sums(z, r_, s_)
        # finish in the exponent-free way: C = s*A + r*B1 + L' + H with the big-int oracle
        from oracle.pyref.curves import Group
        G1, G2 = Group(OC[curve], 1), Group(OC[curve], 2)
        f = c.fq_limbs
        A = jac_limbs_to_affine_oracle(curve, 1, sums[0:3 * f])
        B1 = jac_limbs_to_affine_oracle(curve, 1, sums[3 * f:6 * f])
        B2 = jac_limbs_to_affine_oracle(curve, 2, sums[6 * f:12 * f])
        Hh = jac_limbs_to_affine_oracle(curve, 1, sums[12 * f:15 * f])
        L = jac_limbs_to_affine_oracle(curve, 1, sums[15 * f:18 * f])
        Cc = G1.add(G1.add(G1.mul(A, s_), G1.mul(B1, r_)), G1.add(L, Hh))
        assert (A, B2, Cc) == (unpt(e["a"], 1), unpt(e["b"], 2), unpt(e["c"], 1))
        # every rank holds the same folded sums
        allr = all_gather_points(sums, world)
        assert all(np.array_equal(allr[0], allr[k]) for k in range(world))
        q.put((rank, "ok"))
    except Exception as ex:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_shard_bounds_tile():
    from ckb_zkp_amd.distributed import shard_bounds
    for n in (0, 1, 7, 8, 1258289, (1 << 24) - 1):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_sharded_msm_and_groth16_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
