"""Marlin prover at scale on one MI355X (BASELINE.json configs[3]): MiMC-chain R1CS with ~2^k constraints over BN254,
device-side indexer + device-resident prover, proof checked by the oracle's verifier (AHP equality checks + KZG10
pairing checks) against index commitments computed on the device.

    python tools/marlin_bench.py --log-n 20 --reps 3 > gpurun_out/marlin.json
"""
import argparse
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ckb_zkp_amd import kzg10, marlin_dev  # noqa: E402
from ckb_zkp_amd.api import Context  # noqa: E402
from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain  # noqa: E402
from ckb_zkp_amd.params import get_curve  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--curve", default="bn254")
    ap.add_argument("--samples", type=int, default=0, help="MiMC samples S (default floor((2^k - 1)/10)); 87381 gives |H| = 2^20")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--threads", type=int, default=1, help="prover threads (one context each, sharing index and SRS tables): throughput mode")
    a = ap.parse_args()
    c = get_curve(a.curve)
    log = lambda *m: print("[marlin]", *m, file=sys.stderr, flush=True)
    t = time.perf_counter()
    inst = mimc_chain_instance(a.curve, a.samples or samples_for_domain(a.log_n), seed=0x4D41524C)
    log(f"instance: constraints={inst.num_constraints()} variables={inst.num_inputs + inst.num_aux} ({time.perf_counter() - t:.1f}s)")
    ctx = Context(0)
    t = time.perf_counter()
    didx = marlin_dev.DeviceIndex.from_instance(ctx, inst)
    ctx.sync()
    t_index = time.perf_counter() - t
    log(f"index: |H|={didx.hs} |K|={didx.ks} |B|={didx.bs} max_degree={didx.max_degree} ({t_index:.1f}s)")
    t = time.perf_counter()
    beta_srs = 0x1F2E3D4C5B6A79880102030405060708
    ck = kzg10.setup(ctx, a.curve, didx.max_degree, beta_srs)
    log(f"SRS (trapdoor, device fixed-base) + window tables: {time.perf_counter() - t:.1f}s")
    rnd = random.Random(2026)
    R = dict(w=[rnd.randrange(c.r)], z_a=[rnd.randrange(c.r)], z_b=[rnd.randrange(c.r)],
             mask=[rnd.randrange(c.r) for _ in range(3 * didx.hs)],
             blind={l: [rnd.randrange(c.r), rnd.randrange(c.r)] for l in ("w", "z_a", "z_b", "g_1")},
             blind_shifted={"g_1": [rnd.randrange(c.r), rnd.randrange(c.r)]})
    ch = dict(alpha=rnd.randrange(c.r), eta_a=rnd.randrange(c.r), eta_b=rnd.randrange(c.r), eta_c=rnd.randrange(c.r),
              beta=rnd.randrange(c.r), gamma=rnd.randrange(c.r), xi=rnd.randrange(1 << 128))
    from ckb_zkp_amd import codec
    w_mont = codec.fr_to_mont(inst.z[1:], c).reshape(-1, 4)
    R["mask"] = codec.fr_to_mont(R["mask"], c).reshape(-1, 4)      # sampled once, outside the timed region
    runs = []
    proof = None
    ic = didx.commit_index(ctx, ck)
    ivk = marlin_dev.index_verifier_key(didx, ck, ic, ck.vk_g2)
    for i in range(a.reps):
        tm = {}
        t = time.perf_counter()
        # create_random_proof: verifier messages derived from the Fiat-Shamir transcript round by round
        proof = marlin_dev.create_random_proof(ctx, didx, ck, ivk, (inst.z[:1], w_mont), R, tm)
        tm["wall_s"] = time.perf_counter() - t
        log(f"rep {i}: " + " ".join(f"{k}={v:.3f}" for k, v in tm.items()))
        runs.append(tm)
    throughput = None
    if a.threads > 1:
        # throughput mode: T prover threads, one context (streams + scratch) each; the index vectors and the SRS window
        # tables are shared (zkp_bases_share); ctypes releases the GIL during the calls, the GPU overlaps the proofs
        import threading
        from dataclasses import replace
        ctxs = [ctx] + [Context(0) for _ in range(a.threads - 1)]
        cks = [ck] + [replace(ck, powers_of_g=ck.powers_of_g.share_with(cx), powers_of_gamma_g=ck.powers_of_gamma_g.share_with(cx))
                      for cx in ctxs[1:]]
        per_thread = max(2, a.reps)
        results = [None] * a.threads

        def worker(i):
            for _ in range(per_thread):
                results[i] = marlin_dev.create_proof(ctxs[i], didx, cks[i], (inst.z[:1], w_mont), R, ch)

        for warm in (True, False):
            ths = [threading.Thread(target=worker, args=(i,)) for i in range(a.threads)]
            t = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            dt = time.perf_counter() - t
        throughput = a.threads * per_thread / dt
        same = all(r["commitments"] == proof["commitments"] and r["evaluations"] == proof["evaluations"] and
                   r["opening_proofs"] == proof["opening_proofs"] for r in results)
        log(f"throughput mode: {a.threads} threads x {per_thread} proofs in {dt:.3f}s = {throughput:.2f} proofs/s; identical proofs: {same}")
        assert same
    verified = None
    if not a.no_verify:
        t = time.perf_counter()
        from oracle.pyref import marlin as om
        from oracle.pyref.curves import Group
        from oracle.pyref.ntt import Domain
        from tests.util import OC
        oc = OC[a.curve]
        G1, G2 = Group(oc, 1), Group(oc, 2)
        pp = dict(curve=oc, g=G1.gen, gamma_g=G1.mul(G1.gen, 7), h=G2.gen, beta_h=G2.mul(G2.gen, beta_srs))
        oidx = dict(curve=oc, dh=Domain(oc, didx.hs), dk=Domain(oc, didx.ks), max_degree=didx.max_degree,
                    num_variables=didx.nrows, num_constraints=didx.nrows, num_non_zeros=didx.num_non_zeros)
        wire = dict(commitments=proof["commitments"], evaluations=proof["evaluations"], opening_proofs=proof["opening_proofs"])
        verified = bool(om.verify_random_proof(oidx, pp, ic, wire, []))
        bad = dict(wire, evaluations=[(wire["evaluations"][0] + 1) % c.r] + wire["evaluations"][1:])
        rejected = not om.verify_random_proof(oidx, pp, ic, bad, [])
        log(f"oracle verifier: accept={verified} tampered_rejected={rejected} ({time.perf_counter() - t:.1f}s)")
        verified = verified and rejected
    best = min(runs, key=lambda r: r["total_s"])
    print(json.dumps({"workload": f"Marlin prove, MiMC chain {inst.num_constraints()} constraints, {a.curve}, 1xMI355X",
                      "domain_h": didx.hs, "domain_k": didx.ks, "domain_b": didx.bs, "max_degree": didx.max_degree,
                      "index_s": round(t_index, 3), "prove_s": round(best["total_s"], 4),
                      "proofs_per_s": round(1.0 / best["total_s"], 4),
                      "throughput_proofs_per_s": None if throughput is None else round(throughput, 3), "threads": a.threads,
                      "breakdown_s": {k: round(v, 4) for k, v in best.items()}, "runs": len(runs),
                      "verified_by_oracle_verifier": verified,
                      "note": "create_random_proof: verifier messages derived from the merlin/ChaCha20 Fiat-Shamir transcript round by "
                              "round (commit -> absorb -> squeeze); zk randomness (masks, blinders) sampled outside the timed region"}))


if __name__ == "__main__":
    main()
