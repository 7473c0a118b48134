"""Marlin (BASELINE.json configs[3]) in THROUGHPUT mode on one MI355X: T prover threads, one context + one resident index each (the
contract of include/zkp_accel.h: one ctx per prover thread), the SRS window tables shared (zkp_bases_share), independent proofs
(different zk randomness) — the device fills one proof's Fiat-Shamir waits and reduction tails with the other proofs' kernels.

    python tools/marlin_throughput.py --threads 1,2,3 --proofs 6 > gpurun_out/marlin_tp.json
"""
import argparse
import json
import os
import random
import sys
import threading
import time
from dataclasses import replace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ckb_zkp_amd import codec, kzg10  # noqa: E402
from ckb_zkp_amd import marlin as M  # noqa: E402
from ckb_zkp_amd.api import Context  # noqa: E402
from ckb_zkp_amd.circuits import mimc_chain_instance  # noqa: E402
from ckb_zkp_amd.params import get_curve  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=87381)
    ap.add_argument("--threads", default="1,2,3")
    ap.add_argument("--proofs", type=int, default=6, help="proofs per thread in the timed region")
    ap.add_argument("--stream-offset", type=int, default=0,
                    help="throwaway HIP streams created between two contexts: shifts the hardware queues (stream index mod GPU_MAX_HW_QUEUES) "
                         "the next context's lane-0 streams land on")
    a = ap.parse_args()
    curve = "bn254"
    c = get_curve(curve)
    log = lambda *m: print("[marlin-tp]", *m, file=sys.stderr, flush=True)
    inst = mimc_chain_instance(curve, a.samples, seed=0x4D41524C)
    tmax = max(int(t) for t in a.threads.split(","))
    ctxs = []
    keep_streams = []
    for t in range(tmax):
        if t and a.stream_offset:
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            for _ in range(a.stream_offset):
                st = ctypes.c_void_p()
                assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
                keep_streams.append(st)
        ctxs.append(Context(0))
    idxs = [M.NativeIndex(cx, inst) for cx in ctxs]
    ck0 = kzg10.setup(ctxs[0], curve, idxs[0].max_degree, 0x1F2E3D4C5B6A79880102030405060708)
    cks = [ck0] + [replace(ck0, powers_of_g=ck0.powers_of_g.share_with(cx), powers_of_gamma_g=ck0.powers_of_gamma_g.share_with(cx))
                   for cx in ctxs[1:]]
    ic = idxs[0].commit_index(ck0)
    ivk = M.index_verifier_key(idxs[0], ck0, ic, ck0.vk_g2)
    w_mont = codec.fr_to_mont(inst.z[1:], c).reshape(-1, 4)
    rnd = random.Random(7)
    Rs = []
    for t in range(tmax):
        mask = codec.fr_to_mont([rnd.randrange(c.r) for _ in range(3 * idxs[0].hs)], c).reshape(-1, 4)
        Rs.append(dict(w=[rnd.randrange(c.r)], z_a=[rnd.randrange(c.r)], z_b=[rnd.randrange(c.r)], mask=None,
                       mask_dev=ctxs[t].to_device(mask),
                       blind={l: [rnd.randrange(c.r), rnd.randrange(c.r)] for l in ("w", "z_a", "z_b", "g_1")},
                       blind_shifted={"g_1": [rnd.randrange(c.r), rnd.randrange(c.r)]}))
    # reference proofs (sequential, one context at a time) for the equality check of the concurrent runs
    ref = [M.prove_native(ctxs[t], idxs[t], cks[t], ivk, inst.z[:1], w_mont, Rs[t]) for t in range(tmax)]
    out = {}
    for T in [int(t) for t in a.threads.split(",")]:
        res = [None] * T

        def worker(i, n):
            for _ in range(n):
                res[i] = M.prove_native(ctxs[i], idxs[i], cks[i], ivk, inst.z[:1], w_mont, Rs[i])

        for n in (2, a.proofs):                              # warm, timed
            ths = [threading.Thread(target=worker, args=(i, n)) for i in range(T)]
            t0 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            dt = time.perf_counter() - t0
        same = all(res[i]["commitments"] == ref[i]["commitments"] and res[i]["evaluations"] == ref[i]["evaluations"] and
                   res[i]["opening_proofs"] == ref[i]["opening_proofs"] for i in range(T))
        out[str(T)] = {"threads": T, "proofs": T * a.proofs, "seconds": round(dt, 4), "proofs_per_s": round(T * a.proofs / dt, 3),
                       "ms_per_proof_wall": round(dt / a.proofs * 1e3, 2), "equal_to_sequential": same}
        log(T, out[str(T)])
    print(json.dumps({"stream_offset": a.stream_offset, "workload": f"Marlin create_random_proof, {inst.num_constraints()} constraints, bn254, T contexts x threads on one MI355X",
                      "results": out}))


if __name__ == "__main__":
    main()
