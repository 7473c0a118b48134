#!/usr/bin/env python3
"""Benchmark: Groth16 proofs/sec on a 2^20-constraint MiMC-chain R1CS over BN254 (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: one complete Groth16 proof
(sparse A/B·z, 4 NTTs + pointwise on the evaluation-form key — the reference's witness_map is 7 NTTs; the last transform and the C
chain live in the transformed H / L queries, same proof bytes —, five MSMs incl. the G2 one, assembly -> 3 affine points)
with the proving key, the circuit matrices and the witness z already resident in HBM.  Multi-GPU: one process
per GPU, every rank proves independent proofs with the same resident key (weak scaling, no data-path
collective); `value` = proofs of all ranks / max-over-ranks time.

The JSON line also carries
  roofline      the dominant kernel (MSM bucket accumulation): algorithmic HBM bytes per launch / HIP-event time
  cpu_baseline  the C++ restatement of the reference algorithm (oracle/cpu, "port") on the host cores: the full 2^20 instance
                when the host proves it in seconds (the GPU box: 256 threads, 2.8 s), else a 2^16 sample scaled linearly
                (rank 0, N=1 only); the device proof and witness map of the same (r, s) are compared with the port's.
  hbm_peak_measured  an on-box streaming copy kernel (zkp_bench_hbm_copy) next to the nominal 8 TB/s
and, at N=1 on the default BN254 2^20 line, one block per further BASELINE config, each with its own value, roofline, valu_roof,
cpu_baseline and full-instance parity_check:  marlin_config4 (configs[3]), bls12_381_2p22 (configs[2]),
bn254_2p24_single_gpu (configs[4] on one GPU), bn254_2p20_skewed_witness (50 % of aux in {0,1}).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# must precede the first HIP call (torch.cuda): see ckb_zkp_amd/__init__.py
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

TOXIC = dict(alpha=0x1234567890ABCDEF1, beta=0xFEDCBA09876543211, gamma=0x1111111111111111111,
             delta=0x2222222222222222223, tau=0x3333333333333333335)


DETAIL_PATH = os.environ.get("ZKP_BENCH_DETAIL") or os.path.join("gpurun_out", "bench_detail.json")
LINE_CAP = 4096                      # the driver's parser lost the 21 KB line of round 5: the final stdout line stays under 4 KB


def _short(v, n):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 1] + "~"


def _pick(d, keys, strcap=160):
    return {k: _short(d[k], strcap) for k in keys if isinstance(d, dict) and k in d} if isinstance(d, dict) else None


def compact_line(out):
    """The ONE stdout line of a bench run (<= LINE_CAP bytes): the contract keys, `roofline`, `cpu_baseline`, both halves of
    BASELINE.json's metric (`value` and `msm_g1.mops`) and the `summary` digest.  Everything else (per-config blocks, phase
    timings, notes) goes to DETAIL_PATH, named in `detail`."""
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline")}
    line["dtype"] = out.get("dtype")
    line["data"] = _short(out.get("data"), 60)
    cfg = out.get("config") or {}
    line["config"] = {"workload": _short(cfg.get("workload"), 200), **_pick(cfg, ("curve", "log_domain", "parallelism", "key_upload_s"), 90)}
    if cfg.get("key_form"):
        line["config"]["key_form"] = _short(cfg["key_form"], 150)
    rf = out.get("roofline")
    line["roofline"] = (dict(_pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_kind", "traffic_source",
                                         "avg_launch_ms", "algorithmic_bytes_per_launch", "algorithmic_bytes", "ms",
                                         "peak_measured", "frac_of_measured_peak", "from")),
                             kernel=_short(rf.get("kernel"), 70)) if isinstance(rf, dict) else None)
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = (dict(_pick(cb, ("value", "unit", "cores", "kind", "cpu_model", "s_per_proof", "from")),
                                 sample=_short(cb.get("sample"), 170)) if isinstance(cb, dict) else None)
    if isinstance(out.get("msm_g1"), dict):
        line["msm_g1"] = _pick(out["msm_g1"], ("n", "ms", "mops", "best_ms", "batched_ms", "batched_mops"))
    if isinstance(out.get("parity_check"), dict):
        line["parity_check"] = _pick(out["parity_check"], ("device_eq_cpu_port", "witness_map_eq_cpu_port", "instance"), 60)
    if out.get("scale_parity") is not None:
        line["scale_parity"] = _pick(out["scale_parity"], ("all_ranks_eq_rank0", "sharded_eq_single_gpu", "all_devices_eq_device0", "ranks"))
    if isinstance(out.get("exchange"), dict):
        line["exchange"] = _pick(out["exchange"], ("kind", "ranks", "devices", "bytes_per_rank", "exchange", "rccl_ranks", "witness_map",
                                                   "ms_replicated", "ms_split"), 80)
    if out.get("summary") is not None:
        line["summary"] = out["summary"]
    line["detail"] = out.get("detail")
    txt = json.dumps(line)
    if len(txt) >= LINE_CAP:                                # never lose the line: drop the digest before the contract keys
        for k in ("summary", "exchange", "parity_check"):
            line.pop(k, None)
            txt = json.dumps(line)
            if len(txt) < LINE_CAP:
                break
    assert len(txt) < LINE_CAP, len(txt)
    return txt


def emit(out):
    """full record -> gpurun_out/bench_detail.json (merged back by gpurun; copied into profiles/ per round), compact line -> stdout"""
    try:
        os.makedirs(os.path.dirname(DETAIL_PATH), exist_ok=True)
        with open(DETAIL_PATH, "w") as f:
            json.dump(out, f)
        out["detail"] = DETAIL_PATH
    except OSError as e:
        out["detail"] = f"not written ({e.__class__.__name__})"
    log(f"full record ({len(json.dumps(out))} bytes): {out['detail']}")
    print(compact_line(out), flush=True)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def bench_sharded(args, ctx, c, inst, params, z, rank, world, local):
    """BASELINE configs[4]: one proof per step; every query is split by index over the ranks (each rank keeps 1/N of the
    key resident), the partial MSM results are all-gathered (device tensors, RCCL) and folded + assembled on the device;
    the NTTs are replicated.  Nothing but the 3 proof points leaves HBM inside a step."""
    import torch
    from ckb_zkp_amd.distributed import DeviceShardedGroth16Prover
    device = torch.device("cuda", local)
    prover = DeviceShardedGroth16Prover(ctx, params, inst, rank, world, device=device,
                                        transport="gloo" if args.single_device_test else "nccl")
    r_, s_ = 0x1234567, 0x7654321

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    z_dev = ctx.to_device(z)                                    # witness resident in HBM before the timed region
    # scale_parity, checked BEFORE timing: the sharded proof of a fixed (r, s) == the proof rank 0 computes alone with the whole key
    sharded = prover.prove(z_dev, r_, s_)                       # collective: every rank takes part
    scale_parity = roofline = None
    if rank == 0:
        from ckb_zkp_amd import groth16
        pk0 = groth16.ProvingKey(ctx, params, inst)
        single = pk0.prove_raw(z_dev, codec_mont(c, r_), codec_mont(c, s_), z_on_device=True)
        # per-device roofline of the dominant kernel, measured on rank 0 with the whole key before it is freed (accumulate is
        # per-point work: a shard runs the same kernel over 1/world of the points)
        try:
            roofline, _ = accumulate_roofline(ctx, pk0, c, inst, lambda: pk0.prove_raw(z_dev, codec_mont(c, r_), codec_mont(c, s_), z_on_device=True), None)
            roofline["from"] = "rank 0, whole key on one device (same kernel a shard runs over 1/world of the points)"
        except Exception as e:
            log("roofline on rank 0 failed:", repr(e))
        pk0.free()
        scale_parity = {"sharded_eq_single_gpu": bool(np.array_equal(sharded[0], single[0]) and list(sharded[1]) == list(single[1])),
                        "ranks": world, "note": "one proof with a fixed (r, s): all ranks sharded vs rank 0 alone with the full key"}
        log(f"scale_parity: {scale_parity['sharded_eq_single_gpu']}")

    def step():
        # partial MSMs (this rank's 1/world of every query) -> all_gather_into_tensor on device buffers (RCCL) ->
        # fold + assembly on the device; only the 3 affine proof points come back to the host
        return prover.prove(z_dev, r_, s_)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.single_device_test else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        emit({
            "metric": f"Groth16 proofs/sec (2^{args.log_n} domain, {c.name}), ONE proof base-sharded over the GPUs",
            "value": round(args.steps / dt, 4), "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32 limbs (256-bit Montgomery integers)",
            "data": "synthetic (MiMC-chain R1CS, PRF witness, proving key generated from a fixed trapdoor)",
            "config": {"workload": f"Groth16 prove, MiMC-chain R1CS, {inst.num_constraints()} constraints, {c.name}, "
                                   f"queries sharded {world}-way by index, all-gather of 5 partial points + fold",
                       "curve": c.name, "log_domain": args.log_n, "parallelism": f"base-sharded x{world}",
                       "note": "witness, h, partial sums and the gathered buffer resident in HBM; NTT pipeline replicated per rank; "
                               "per step: zkp_groth16_prove_partials_dev -> all_gather_into_tensor -> zkp_groth16_fold_assemble_dev"},
            "scale_parity": scale_parity,
            "exchange": {"kind": "gloo (host-staged, TEST ONLY)" if args.single_device_test else "rccl" if world > 1 else "none",
                         "ranks": world, "call": "torch.distributed.all_gather_into_tensor on device buffers" if world > 1 else None,
                         "bytes_per_rank": prover.pb},
            "roofline": roofline, "cpu_baseline": recorded_cpu_baseline(c.name, args.log_n)})
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def codec_mont(c, v):
    from ckb_zkp_amd import codec
    return codec.fr_to_mont([v], c)[0]


def bench_single_process_multi(args):
    """`python bench.py --gpus N` WITHOUT torchrun: one process drives N GPUs through the library's own multi-GPU entry points
    (include/zkp_accel.h: zkp_ctx_create_multi + zkp_groth16_prove_batch_multi / zkp_groth16_prove_multi) — what a Rust caller
    of create_proof gets.  throughput: independent proofs round-robin over replicated keys (weak scaling); shard: ONE proof per
    step, every query base-sharded, partial sums gathered over xGMI inside the library (BASELINE configs[4])."""
    from ckb_zkp_amd import codec, groth16
    from ckb_zkp_amd.api import MultiContext
    from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain
    from ckb_zkp_amd.params import get_curve
    ids = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    n = len(ids)
    c = get_curve(args.curve)
    m = MultiContext(ids)
    inst = mimc_chain_instance(c, samples_for_domain(args.log_n))
    t0 = time.time()
    params = groth16.generate_parameters(m, c, inst, **TOXIC)
    log(f"instance + synthetic key: {time.time()-t0:.1f}s; devices {ids}")
    shard = args.mode == "shard"
    t0 = time.time()
    mpk = groth16.MultiProvingKey(m, params, inst, groth16.MULTI_SHARD if shard else groth16.MULTI_REPLICATE)
    log(f"key upload on {n} device(s) ({'sharded' if shard else 'replicated'}): {time.time()-t0:.1f}s")
    z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
    zds = [m.member(k).to_device(z) for k in range(n)]          # witness resident in HBM on every device before the timed region
    rng = np.random.default_rng(1234)

    def rand_fr(k):
        return np.stack([codec.fr_to_mont([int.from_bytes(rng.bytes(32), "little") % c.r], c)[0] for _ in range(k)])

    def steps(k):
        if k <= 0:
            return
        if shard:
            for _ in range(k):
                mpk.prove_raw(zds, rand_fr(1)[0], rand_fr(1)[0], z_on_device=True)
        else:
            # `k` steps = k proofs PER GPU, like one rank per GPU in the torchrun mode
            mpk.prove_batch_raw([zds[i % n] for i in range(k * n)], rand_fr(k * n), rand_fr(k * n), z_on_device=True)

    def sync_all():
        for k in range(n):
            m.member(k).sync()

    # scale_parity, checked BEFORE timing, with a fixed (r, s): shard — the proof over all devices == the proof device 0 computes alone
    # with the whole key; throughput — the same proof from every device's replica
    r_fix, s_fix = rand_fr(1)[0], rand_fr(1)[0]
    if shard:
        multi = mpk.prove_raw(zds, r_fix, s_fix, z_on_device=True)
        pk0 = groth16.ProvingKey(m.member(0), params, inst)
        single = pk0.prove_raw(zds[0], r_fix, s_fix, z_on_device=True)
        pk0.free()
        ok = bool(np.array_equal(multi[0], single[0]) and list(multi[1]) == list(single[1]))
        scale_parity = {"sharded_eq_single_gpu": ok, "ranks": n,
                        "note": "zkp_groth16_prove_multi over all devices vs zkp_groth16_prove_dev on device 0 with the full key"}
    else:
        outs, infs = mpk.prove_batch_raw(zds, np.stack([r_fix] * n), np.stack([s_fix] * n), z_on_device=True)
        ok = bool(all(np.array_equal(outs[k], outs[0]) and list(infs[k]) == list(infs[0]) for k in range(n)))
        scale_parity = {"all_devices_eq_device0": ok, "ranks": n, "note": "the same (z, r, s) proved once on every device's replica"}
    log(f"scale_parity: {ok}")

    steps(args.warmup)
    sync_all()
    t0 = time.perf_counter()
    steps(args.steps)
    sync_all()
    dt = time.perf_counter() - t0
    proofs = args.steps if shard else args.steps * n
    # per-device roofline of the dominant kernel: device 0 alone with its own copy of the whole key (outside the timed region)
    roofline = None
    try:
        pk0 = groth16.ProvingKey(m.member(0), params, inst)
        roofline, _ = accumulate_roofline(m.member(0), pk0, c, inst, lambda: pk0.prove_raw(zds[0], r_fix, s_fix, z_on_device=True), None)
        roofline["from"] = "device 0 alone, whole key (per-device figure; the same kernel runs on every device)"
        pk0.free()
    except Exception as e:
        log("per-device roofline failed:", repr(e))
    emit({
        "metric": "Groth16 proofs/sec (2^20 constraints, BN256)" if (args.log_n == 20 and c.name == "bn254" and not shard)
        else f"Groth16 proofs/sec (2^{args.log_n} domain, {c.name})" + (", ONE proof base-sharded over the GPUs" if shard else ""),
        "value": round(proofs / dt, 4), "unit": "proofs/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if shard else "weak",
        "vs_baseline": None, "dtype": "u32 limbs (256-bit Montgomery integers)",
        "data": "synthetic (MiMC-chain R1CS, PRF witness, proving key generated from a fixed trapdoor)",
        "config": {"workload": f"Groth16 prove, MiMC-chain R1CS, {inst.num_constraints()} constraints (domain 2^{args.log_n}), {c.name}",
                   "curve": c.name, "log_domain": args.log_n, "devices": ids,
                   "parallelism": (f"base-sharded x{n}, in-library exchange (zkp_groth16_prove_multi)" if shard
                                   else f"independent proofs x{n}, one process, one host thread per GPU (zkp_groth16_prove_batch_multi)"),
                   "launch": "single process (no torchrun): the C-ABI multi-GPU path"},
        "scale_parity": scale_parity,
        "exchange": (dict(mpk.info(), call="inside zkp_groth16_prove_multi: ncclAllGather when the devices are distinct, else "
                                            "hipMemcpyPeerAsync (said on stderr); witness-map variant measured on proofs 3-4 of the key")
                     if shard else {"kind": "none (independent proofs, replicated key)", "devices": n}),
        "roofline": roofline, "cpu_baseline": recorded_cpu_baseline(c.name, args.log_n)})
    for k, d in enumerate(zds):
        m.member(k).dev_free(d)
    mpk.free()
    m.close()


def bench_marlin(ctx, curve="bn254", samples=87381, reps=3, verify=True, cpu=False):
    """BASELINE configs[3]: Marlin create_random_proof (|H| = 2^20, |K| = 2^21, |B| = 2^23, SRS degree 6.29 M) on one GPU:
    device-side indexer + device-resident prover, verifier messages derived from the Fiat-Shamir transcript round by
    round, the proof checked by the oracle's verifier (which re-derives them).  -> dict for the JSON line."""
    import random
    from ckb_zkp_amd import codec, kzg10
    from ckb_zkp_amd import marlin as marlin_dev
    from ckb_zkp_amd.circuits import mimc_chain_instance
    from ckb_zkp_amd.params import get_curve
    c = get_curve(curve)
    inst = mimc_chain_instance(curve, samples, seed=0x4D41524C)
    t = time.perf_counter()
    nidx = marlin_dev.NativeIndex(ctx, inst)                  # zkp_marlin_index_upload: arithmetization computed on the device
    ctx.sync()
    t_index = time.perf_counter() - t
    didx = nidx                                               # (same attribute names: hs, ks, bs, max_degree, nrows, ...)
    beta_srs = 0x1F2E3D4C5B6A79880102030405060708
    ck = kzg10.setup(ctx, curve, nidx.max_degree, beta_srs)
    rnd = random.Random(2026)
    mask = codec.fr_to_mont([rnd.randrange(c.r) for _ in range(3 * nidx.hs)], c).reshape(-1, 4)
    mask_dev = ctx.to_device(mask)                            # zk randomness: sampled and placed before the timed region
    R = dict(w=[rnd.randrange(c.r)], z_a=[rnd.randrange(c.r)], z_b=[rnd.randrange(c.r)], mask=None, mask_dev=mask_dev,
             blind={l: [rnd.randrange(c.r), rnd.randrange(c.r)] for l in ("w", "z_a", "z_b", "g_1")},
             blind_shifted={"g_1": [rnd.randrange(c.r), rnd.randrange(c.r)]})
    w_mont = codec.fr_to_mont(inst.z[1:], c).reshape(-1, 4)
    ic = nidx.commit_index(ck)
    ivk = marlin_dev.index_verifier_key(nidx, ck, ic, ck.vk_g2)
    runs, proof = [], None
    for _ in range(reps + 1):
        t = time.perf_counter()
        proof = marlin_dev.prove_native(ctx, nidx, ck, ivk, inst.z[:1], w_mont, R)     # ONE C call: zkp_marlin_prove
        tm = proof["timing"]
        runs.append({"total_s": time.perf_counter() - t, "rounds_s": sum(tm["ms_round"]) * 1e-3,
                     "commits_s": sum(tm["ms_commit"]) * 1e-3, "evaluations_s": tm["ms_evaluations"] * 1e-3,
                     "batch_open_s": tm["ms_open"] * 1e-3, "timing": tm})
    best = min(runs[1:], key=lambda r_: r_["total_s"])
    tm = best.pop("timing")
    for r_ in runs:
        r_.pop("timing", None)
    # roofline of the dominant phase: the commitment MSMs (zkp_msm_g1_mont_batch_dev inside the call).  Algorithmic bytes =
    # 32 B per coefficient + one 64-B affine SRS power per coefficient, over the time of the three commit phases
    # (which also hold the blinding MSMs, the transcript and the affine conversions: the figure is conservative).
    fq = c.fq_limbs * 8
    commit_bytes = tm["commit_points"] * (32 + 2 * fq)
    # Since round 3 the commitments of the mask polynomial and t(X) start DURING their rounds (marlin.hip, early commitments), so
    # the commit phases alone no longer hold all the MSM time: the clock is rounds + commits (conservative: it also holds the NTTs).
    commit_s = (sum(tm["ms_commit"]) + sum(tm["ms_round"])) * 1e-3
    mpmc, msrc = recorded_latest("pmc_marlin_accumulate.json")
    roofline = {"bound": "hbm", "kernel": "accumulate_kernel inside the commitment MSMs (PC::commit of the three AHP rounds)",
                "achieved": round(commit_bytes / commit_s / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(commit_bytes / commit_s / 1e9 / 8000.0, 5),
                "traffic": mpmc["traffic_bytes_per_proof_commit_msms"] if mpmc else None,
                "traffic_kind": "recorded" if mpmc else None, "traffic_source": msrc,
                "algorithmic_bytes": int(commit_bytes), "ms": round(commit_s * 1e3, 3), "points": tm["commit_points"],
                "ns_per_point": round(commit_s * 1e9 / max(tm["commit_points"], 1), 3),
                "note": "integer-VALU bound like the Groth16 MSMs (DESIGN.md); time = the three AHP rounds + their commit phases "
                        "(early commitments overlap the rounds)"}
    cpu_baseline, parity_check = None, None
    if cpu:
        # CPU port of the WHOLE prover (oracle/cpu/marlin_oracle.inc: marlin::create_random_proof restated phase by phase, transcript
        # by oracle/pyref/fs_rng.py), built from the instance as synthesised and the committer key's host arrays, timed on all host
        # threads on this very instance: no extrapolation.  Its proof is the checker of the device proof (parity_check).
        from oracle import cpu_oracle
        from oracle.pyref import marlin as om
        from oracle.pyref.fields import BN254, BLS12_381
        from oracle.pyref.ntt import Domain
        oc = BN254 if c.name == "bn254" else BLS12_381
        cores = cpu_oracle.hardware_threads()
        t = time.perf_counter()
        co = cpu_oracle.MarlinOracle(oc, inst, srs=(ck.host_g, ck.host_gamma_g), threads=cores)
        t_oidx = time.perf_counter() - t
        oidx_ = dict(curve=oc, dh=Domain(oc, nidx.hs))
        t = time.perf_counter()
        oic = co.index_commitments()
        t_oic = time.perf_counter() - t
        o = co.create_proof(inst.z[:1], w_mont, dict(R, mask=mask), om.FiatShamirChallenger(oidx_, ivk, []))
        lbl = marlin_dev.LABELS_1 + marlin_dev.LABELS_2 + marlin_dev.LABELS_3
        parity_check = {"index_commitments_eq_cpu_port": oic == ic,
                        "challenges_eq_cpu_port": o["challenges"] == proof["challenges"],
                        "commitments_eq_cpu_port": all(o["commitments"][l] == proof["commitments"][l] for l in lbl),
                        "evaluations_eq_cpu_port": o["evaluations"] == proof["evaluations"],
                        "opening_proofs_eq_cpu_port": o["opening_proofs"] == proof["opening_proofs"],
                        "compared": "12 index commitments, 9 + 2 shifted commitments, 21 evaluations, 2 opening proofs (w, rand_v), "
                                    "7 derived verifier messages; full instance"}
        parity_check["device_eq_cpu_port"] = all(v for k_, v in parity_check.items() if k_.endswith("_cpu_port"))
        cpu_baseline = {"value": round(1.0 / o["seconds"], 5), "unit": "proofs/s", "cores": cores, "kind": "port",
                        "s_per_proof": round(o["seconds"], 3), "phase_s": {k_: round(v, 3) for k_, v in o["phase_seconds"].items()},
                        "index_s": round(t_oidx, 3), "index_commit_s": round(t_oic, 3),
                        "sample": f"oracle/cpu Marlin prover (create_random_proof restated in C++) on {cores} threads, the full "
                                  "instance, one run, no extrapolation; a round's commitments run concurrently (more parallel than "
                                  "the reference's sequential PC::commit: favours the CPU)"}
        co.free()
    return_extra = {"roofline": roofline, "cpu_baseline": cpu_baseline, "parity_check": parity_check, "counts": {k_: tm[k_] for k_ in ("commit_points", "open_points", "ntt_count", "ntt_elements")},
                    "phase_ms": {"rounds": [round(x, 3) for x in tm["ms_round"]], "commits": [round(x, 3) for x in tm["ms_commit"]],
                                 "evaluations": round(tm["ms_evaluations"], 3), "batch_open": round(tm["ms_open"], 3)}}
    verified = None
    if verify:
        from oracle.pyref import marlin as om              # the checker (oracle verifier), outside any timed region
        from oracle.pyref.curves import Group
        from oracle.pyref.fields import BN254, BLS12_381
        from oracle.pyref.ntt import Domain
        oc = BN254 if c.name == "bn254" else BLS12_381
        G1, G2 = Group(oc, 1), Group(oc, 2)
        pp = dict(curve=oc, g=G1.gen, gamma_g=G1.mul(G1.gen, 7), h=G2.gen, beta_h=G2.mul(G2.gen, beta_srs))
        oidx = dict(curve=oc, dh=Domain(oc, didx.hs), dk=Domain(oc, didx.ks), max_degree=didx.max_degree,
                    num_variables=didx.nrows, num_constraints=didx.nrows, num_non_zeros=didx.num_non_zeros)
        wire = dict(commitments=proof["commitments"], evaluations=proof["evaluations"], opening_proofs=proof["opening_proofs"])
        bad = dict(wire, evaluations=[(wire["evaluations"][0] + 1) % c.r] + wire["evaluations"][1:])
        verified = bool(om.verify_random_proof(oidx, pp, ic, wire, [])) and not om.verify_random_proof(oidx, pp, ic, bad, [])
    ck.powers_of_g.free()
    ck.powers_of_gamma_g.free()
    ctx.dev_free(mask_dev)
    nidx.free()
    return {"workload": f"Marlin create_random_proof, MiMC chain {inst.num_constraints()} constraints, {c.name}, 1xMI355X "
                        f"(|H|=2^{didx.hs.bit_length()-1}, |K|=2^{didx.ks.bit_length()-1}, |B|=2^{didx.bs.bit_length()-1}, SRS degree {didx.max_degree})",
            "value": round(1.0 / best["total_s"], 3), "unit": "proofs/s", "s_per_proof": round(best["total_s"], 4),
            "breakdown_s": {k: round(v, 4) for k, v in best.items()}, "index_s": round(t_index, 3), "runs": reps,
            **return_extra,
            "verified_by_reference_verifier_restatement": verified,
            "note": "one zkp_marlin_prove call per proof (csrc/marlin.hip: no Python between the rounds): prover_init, AHP round -> "
                    "PC::commit (batched MSMs) -> absorb -> squeeze with the library's merlin/ChaCha20 FiatShamirRng, 21 "
                    "evaluations, batch_open; witness uploaded from the host inside the timed region, zk randomness (mask "
                    "polynomial, blinders) sampled before it"}


def summary_block(out, marlin, extra):
    """Compact digest (<= 1.5 KB) of the whole line, printed as its LAST key: per BASELINE config value / ms / parity / accumulate
    fraction of the integer-VALU ceiling / HBM roofline fraction, both halves of BASELINE.json's metric (proofs/s and MSM G1 Mop/s),
    single-proof latency, the NTT and scalar-scan roofline fractions, Marlin."""
    def g(d, *ks):
        for k in ks:
            if not isinstance(d, dict) or d.get(k) is None:
                return None
            d = d[k]
        return d

    def cfg(b):
        if not isinstance(b, dict) or "value" not in b:
            return {"error": (b or {}).get("error") or (b or {}).get("skipped")} if isinstance(b, dict) else None
        return {"v": b["value"], "ms": b.get("ms_per_step"), "par": g(b, "parity_check", "device_eq_cpu_port"),
                "acc": [g(b, "valu_roof", "g1_accumulate", "frac"), g(b, "valu_roof", "g2_accumulate", "frac")], "roof": g(b, "roofline", "frac"),
                "cpu_s": g(b, "cpu_baseline", "s_per_proof")}

    sm = {"unit": "proofs/s", "bn254_2p20": cfg(out), "msm_g1_mops": g(out, "msm_g1", "mops"),
          "msm_g1_batched_mops": g(out, "msm_g1", "batched_mops"), "latency_ms": g(out, "latency", "ms_per_proof"),
          "ntt": {"frac": g(out, "roofline_ntt", "frac"), "ms": g(out, "roofline_ntt", "ms_per_transform"),
                  "valu": g(out, "valu_roof", "ntt", "frac")},
          "scan_frac": g(out, "roofline_scan", "frac"), "h2d_v": g(out, "with_h2d", "value"),
          "hbm_copy_gbs": g(out, "hbm_peak_measured", "value")}
    for short, key in (("skew", "bn254_2p20_skewed_witness"), ("bls22", "bls12_381_2p22"), ("bn24", "bn254_2p24_single_gpu")):
        if key in extra:
            sm[short] = cfg(extra[key])
    if isinstance(marlin, dict):
        sm["marlin"] = ({"v": marlin["value"], "ms": round(marlin["s_per_proof"] * 1e3, 2), "par": g(marlin, "parity_check", "device_eq_cpu_port"),
                         "verified": marlin.get("verified_by_reference_verifier_restatement"), "roof": g(marlin, "roofline", "frac"),
                         "cpu_s": g(marlin, "cpu_baseline", "s_per_proof")} if "value" in marlin else {"error": marlin.get("error")})
    sm["cpu_cores"] = g(out, "cpu_baseline", "cores")
    sm["wall_s"] = out.get("bench_wall_s")
    return sm


def cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def make_rand_fr(c, seed):
    from ckb_zkp_amd import codec
    rng = np.random.default_rng(seed)

    def rand_fr():
        v = int.from_bytes(rng.bytes(32), "little") % c.r
        return codec.fr_to_mont([v], c)[0]
    return rand_fr


def hbm_peak_block(ctx):
    """the on-box streaming-copy bandwidth next to the nominal 8 TB/s (zkp_bench_hbm_copy, 2 x 2 GiB buffers)."""
    g = max(ctx.bench_hbm_copy(1 << 30), ctx.bench_hbm_copy(2 << 30))
    return {"value": round(g, 1), "unit": "GB/s", "nominal": 8000.0, "frac_of_nominal": round(g / 8000.0, 4),
            "note": "zkp_bench_hbm_copy: grid-stride dwordx4 copy kernels (1 / 4 loads in flight, plain / non-temporal, 4-32 workgroups "
                    "per CU) between two 1 GiB and two 2 GiB buffers, read + written bytes / HIP-event time, best variant, measured in "
                    "this process; every roofline block quotes its fraction of BOTH peaks"}


def recorded_latest(suffix):
    """the newest profiles/rNN_<suffix> a round has recorded"""
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        d, src = recorded(f"{rnd}_{suffix}", quiet=True)
        if d:
            return d, src
    log(f"no recorded profiles/rNN_{suffix}: the fields that quote it stay null")
    return None, None


def recorded(name, quiet=False):
    """a value recorded under profiles/ by a separate rocprofv3 --pmc pass (PMC counters cannot be read from inside the process)"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
    try:
        return json.load(open(path)), "profiles/" + name
    except (OSError, ValueError) as e:
        if not quiet:
            log(f"recorded PMC source profiles/{name} unavailable ({e.__class__.__name__}): the fields that quote it stay null")
        return None, None


def recorded_cpu_baseline(curve, log_n):
    """N > 1 lines: the CPU port is timed on rank 0 at N = 1 only (contract); the multi-GPU lines quote that run's block"""
    for rnd in ("r06", "r05", "r04"):
        d, src = recorded(f"{rnd}_bench_bn254_2p20.json", quiet=True)
        if not d:
            continue
        blk = (d.get("cpu_baseline") if (curve, log_n) == ("bn254", 20) else
               (d.get("bn254_2p24_single_gpu") or {}).get("cpu_baseline") if (curve, log_n) == ("bn254", 24) else
               (d.get("bls12_381_2p22") or {}).get("cpu_baseline") if (curve, log_n) == ("bls12_381", 22) else None)
        if isinstance(blk, dict):
            return dict({k: blk[k] for k in ("value", "unit", "cores", "kind", "cpu_model", "s_per_proof", "sample") if k in blk},
                        **{"from": f"{src} (the N=1 run of this config on the same box type: the CPU port does not depend on the GPU count)"})
    return None


def with_measured(block, hbm_meas):
    """adds the fraction of the MEASURED copy peak next to `frac` (of the nominal 8 TB/s)"""
    if block and hbm_meas and block.get("achieved") is not None:
        block["peak_measured"] = hbm_meas["value"]
        block["frac_of_measured_peak"] = round(block["achieved"] / hbm_meas["value"], 5)
    return block


def accumulate_roofline(ctx, pk, c, inst, step, hbm_meas, traffic=None, traffic_src=None):
    """dominant kernel (bucket accumulation): algorithmic HBM bytes per launch / HIP-event time, measured live on the library's
    streams (zkp_set_profiling).  -> (roofline dict, last profiled phase timings)"""
    ctx.set_profiling(True)
    acc_ms, launches, tm_last = 0.0, 0, None
    for _ in range(2):
        step()
        tm_last = pk.last_timing()
        acc_ms += tm_last["ms_msm_accumulate"]
        launches += tm_last["msm_accumulate_launches"]
    ctx.set_profiling(False)
    fq = c.fq_limbs * 8
    nz = inst.num_inputs + inst.num_aux
    # algorithmic bytes: scalars read once (32 B) + affine bases read once, per MSM
    msm_bytes = [(nz + 4) * (32 + 2 * fq), (nz + 4) * (32 + 2 * fq), (nz + 4) * (32 + 4 * fq),
                 (pk.domain_size - 1) * (32 + 2 * fq), (inst.num_aux + 4) * (32 + 2 * fq)]
    bytes_per_launch = sum(msm_bytes) / max(tm_last["msm_accumulate_launches"], 1)
    avg_ms = acc_ms / max(launches, 1)
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": "accumulate_kernel (MSM bucket accumulation, 5 launches/proof)",
                "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 5),
                "traffic": traffic, "traffic_kind": "recorded" if traffic is not None else None, "traffic_source": traffic_src,
                "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(bytes_per_launch),
                "note": "integer-VALU bound (DESIGN.md).  traffic is RECORDED, not measured by this run: PMC counters cannot be read "
                        "from inside the process, the figure is FETCH_SIZE*2 + WRITE_SIZE per launch from the separate rocprofv3 "
                        "--pmc passes of this same command committed under profiles/ (the window-table design gathers each base "
                        "W=13 times at 128-B fabric granularity)"}
    return with_measured(roofline, hbm_meas), tm_last


def valu_roof_block(ctx, c, phases, ntt_gmul=None):
    """the binding roof: integer VALU.  Montgomery products/s sustained by the accumulate kernels vs the rate of the same
    multipliers in a pure multiply loop on every CU (zkp_bench_mulmod, measured now)"""
    ceil_u = ctx.bench_mulmod(c, 1, True)
    ceil_s = ctx.bench_mulmod(c, 1, False)
    ceil_fr = ctx.bench_mulmod(c, 0, True)                # the NTT products run on the unsaturated multiplier since round 3
    g1 = [0, 1, 3, 4]
    t_g1 = sum(phases["ms_msm_acc"][i] for i in g1)
    e_g1 = sum(phases["msm_entries"][i] for i in g1)
    t_g2, e_g2 = phases["ms_msm_acc"][2], phases["msm_entries"][2]
    # product-equivalents of multiplier work (L*L partial products + L*L reduction products = one Montgomery product):
    # G1 madd-2008-s with y3 as one lazily reduced sum of two products: 9.5; G2 with schoolbook Fq2 products as
    # lazily reduced sums (two reductions per Fq2 product, one per component of y3): 27.6  (MSM section)
    a1 = 9.5 * e_g1 / (t_g1 * 1e-3) / 1e9 if t_g1 > 0 else 0.0
    a2 = 27.6 * e_g2 / (t_g2 * 1e-3) / 1e9 if t_g2 > 0 else 0.0
    out = {"unit": "1e9 Montgomery products/s",
           "g1_accumulate": {"achieved": round(a1, 1), "ceiling": round(ceil_u, 1), "frac": round(a1 / ceil_u, 3),
                             "multiplier": "unsaturated 29/28-bit limbs (unsat_dev.hpp)"},
           "g2_accumulate": {"achieved": round(a2, 1), "ceiling": round(ceil_u, 1), "frac": round(a2 / ceil_u, 3),
                             "multiplier": "unsaturated limbs, schoolbook Fq2 with lazily reduced sums (unsat_dev.hpp)"},
           "saturated_fq_ceiling": round(ceil_s, 1), "fr_ceiling": round(ceil_fr, 1),
           "note": "ceilings = zkp_bench_mulmod (better of 2 and 4 independent product chains per lane, 8 workgroups per CU), "
                   "measured in this process; the accumulate / NTT kernels are bound by this roof, not by HBM"}
    if ntt_gmul is not None:
        out["ntt"] = {"achieved": ntt_gmul, "ceiling": round(ceil_fr, 1), "frac": round(ntt_gmul / ceil_fr, 3),
                      "multiplier": "tile unsaturated in LDS (decimation in time, no reductions between stages; ntt.hip)"}
    return out


def ntt_roofline(ctx, c, log_n, hbm_meas):
    """NTT butterfly passes: 64*N algorithmic bytes per transform (read + write once); every pass moves 64*N too"""
    from ckb_zkp_amd.api import NTT_FFT, NTT_COSET_IFFT
    N = 1 << log_n
    buf = ctx.to_device(np.frombuffer(np.random.default_rng(3).bytes(32 * N), dtype=np.uint64).reshape(-1, 4) >> np.uint64(3))
    res = {}
    for name, op in (("fft", NTT_FFT), ("coset_ifft", NTT_COSET_IFFT)):
        for _ in range(3):
            ctx.ntt_dev(c, buf, log_n, op)
        reps = 30
        ctx.timer_start()
        for _ in range(reps):
            ctx.ntt_dev(c, buf, log_n, op)
        res[name] = ctx.timer_stop_ms() / reps
    ctx.dev_free(buf)
    passes = -(-log_n // 7)                                # LDS tiles of <= 2^7 points per pass (ntt.hip)
    t_ntt = res["fft"]
    ach = 64.0 * N / (t_ntt * 1e-3) / 1e9
    mm = 0.5 * log_n + 0.25 * passes      # per pass: S/2 - 3/4 products in the tile (DIT: trivial twiddles fall in the first stages) + 1 at the store
    out = {"bound": "hbm", "kernel": f"NTT pass kernels x{passes} (Stockham passes of one 2^{log_n} transform)",
           "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
           "ms_per_transform": round(t_ntt, 4), "ms_coset_ifft": round(res["coset_ifft"], 4), "algorithmic_bytes": 64 * N,
           "per_pass": {"passes": passes, "ms": round(t_ntt / passes, 4), "achieved": round(ach * passes, 1),
                        "frac": round(ach * passes / 8000.0, 4),
                        "frac_of_measured_peak": round(ach * passes / hbm_meas["value"], 4) if hbm_meas else None,
                        "note": "one pass reads and writes the vector once (64*N bytes)"},
           "valu": {"mulmods_per_element": round(mm, 2), "gmulmod_per_s": round(mm * N / (t_ntt * 1e-3) / 1e9, 1)}}
    pmc, src = recorded_latest("pmc_ntt.json")
    if pmc:
        out["valu_busy_recorded"] = dict(pmc, source=src, kind="recorded",
                                         note="SQ_ACTIVE_INST_VALU x 4 / SQ_BUSY_CYCLES per SIMD of ntt_pass2_kernel (tools/pmc_ntt.sh): the share "
                                              "of the kernel's cycles in which a SIMD's vector ALU is issuing — the binding roof of this kernel")
    return with_measured(out, hbm_meas)


def scan_roofline(phases, hbm_meas):
    """MSM scalar scan (digit extraction fused into the level-1 histogram + scatter passes of the bucket sort)"""
    if not phases or phases.get("ms_msm_scan", 0) <= 0:
        return None
    ach = phases["msm_scan_bytes"] / (phases["ms_msm_scan"] * 1e-3) / 1e9
    return with_measured({"bound": "hbm", "kernel": "sort_hist_kernel + count scan + sort_scatter_kernel (MSM scalar scan)",
                          "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
                          "ms": round(phases["ms_msm_scan"] / phases["msm_scan_launches"], 4),
                          "algorithmic_bytes": int(phases["msm_scan_bytes"] / phases["msm_scan_launches"]),
                          "note": "per MSM: scalars read by the level-1 histogram + scatter passes (32 B each) + 8 B per "
                                  "(bucket, point) entry written"}, hbm_meas)


def cpu_port_block(pk, params, inst, z, c, cores, runs, what, rand_fr):
    """the C++ restatement (oracle/cpu) on the FULL instance: `runs` timed proofs on all host threads, and the device proof +
    witness map of the same (r, s) compared with the port's limb for limb.  -> (cpu_baseline, parity_check)"""
    from oracle import cpu_oracle
    import statistics
    r_fix, s_fix = rand_fr(), rand_fr()
    d_out, d_inf = pk.prove_raw(z, r_fix, s_fix)
    ts, c_out, c_inf, ph, c_h = [], None, None, None, None
    for _ in range(runs):
        t0 = time.perf_counter()
        # (the last run also hands back the quotient h its proof was made from: no separate witness_map pass for the h comparison;
        #  the extra 32 N-byte copy is inside the timed call — it favours the device by < 0.1 %)
        c_out, c_inf, ph, c_h = cpu_oracle.groth16_prove(params, inst, z, r_fix, s_fix, threads=cores, want_h=True)
        ts.append(time.perf_counter() - t0)
    h_eq = bool(np.array_equal(pk.witness_map(z), c_h))
    del c_h
    parity = {"device_eq_cpu_port": bool(np.array_equal(d_out, c_out) and np.array_equal(d_inf, c_inf)),
              "witness_map_eq_cpu_port": h_eq, "instance": what,
              "note": "same (r, s): zkp_groth16_prove vs oracle/cpu groth16_prove, proof limbs + identity flags; "
                      "zkp_groth16_witness_map vs the port's witness_map, all N coefficients"}
    t_all = statistics.median(ts)
    sample = (f"{what} ({inst.num_constraints()} constraints), median of {runs} run(s) {[round(x, 2) for x in ts]} s on {cores} "
              "threads (no extrapolation)")
    base = {"value": round(1.0 / t_all, 6), "unit": "proofs/s", "cores": cores, "kind": "port", "cpu_model": cpu_model_name(),
            "s_per_proof": round(t_all, 3),
            "sample": sample + "; oracle/cpu = C++ restatement of ark-ec/ark-poly 0.2 (Pippenger with the arkworks window rule "
                      "c = ln(n)+2, one thread per window like ark's rayon path: an MSM uses <= ceil(254/c)+1 ~ 17 threads however "
                      "many cores the host has; radix-2 NTT), not the Rust binary",
            "phase_ms": [round(x, 1) for x in ph.tolist()]}
    return base, parity


def config_block(ctx, curve, log_n, steps, warmup, hbm_meas, skewed=False, cpu_runs=1):
    """One further BASELINE config (or the skewed-witness variant) measured like the primary line, on its own key: value,
    ms_per_step, roofline, valu_roof, cpu_baseline and parity_check on the FULL instance.  The key is freed on return."""
    from ckb_zkp_amd import codec, groth16
    from ckb_zkp_amd.circuits import boolean_mimc_instance, mimc_chain_instance, samples_for_domain
    from ckb_zkp_amd.params import get_curve
    from oracle import cpu_oracle
    c = get_curve(curve)
    t_setup = time.time()
    inst = boolean_mimc_instance(c, log_n) if skewed else mimc_chain_instance(c, samples_for_domain(log_n))
    params = groth16.generate_parameters(ctx, c, inst, **TOXIC)
    pk = groth16.ProvingKey(ctx, params, inst)
    z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
    z_dev = ctx.to_device(z)
    t_setup = time.time() - t_setup
    log(f"[{curve} 2^{log_n}{' skewed' if skewed else ''}] instance + key + upload: {t_setup:.1f}s")
    rand_fr = make_rand_fr(c, 4321 + log_n)
    try:
        def batch(k):
            rs = np.stack([rand_fr() for _ in range(k)]), np.stack([rand_fr() for _ in range(k)])
            return lambda: pk.prove_batch_raw([z_dev] * k, rs[0], rs[1])
        batch(warmup)()
        ctx.sync()
        run = batch(steps)
        t0 = time.perf_counter()
        run()
        ctx.sync()
        dt = time.perf_counter() - t0
        step = lambda: pk.prove_raw(z_dev, rand_fr(), rand_fr(), z_on_device=True)
        step()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        lat = (time.perf_counter() - t0) / 3
        roofline, phases = accumulate_roofline(ctx, pk, c, inst, step, hbm_meas)
        valu = valu_roof_block(ctx, c, phases)
        cores = cpu_oracle.hardware_threads()
        what = f"full 2^{log_n}-domain {'boolean-heavy (50 % of aux in {0,1}) ' if skewed else ''}instance"
        cpu_baseline, parity = cpu_port_block(pk, params, inst, z, c, cores, cpu_runs, what, rand_fr)
        out = {"workload": f"Groth16 prove, {'MiMC chain + booleans' if skewed else 'MiMC-chain'} R1CS, {inst.num_constraints()} "
                           f"constraints (domain 2^{log_n}), {inst.num_aux} aux, {c.name}, 1xMI355X",
               "value": round(steps / dt, 4), "unit": "proofs/s", "steps": steps, "warmup": warmup,
               "ms_per_step": round(dt / steps * 1e3, 3), "latency_ms": round(lat * 1e3, 3),
               "table_plan": pk.table_plan(), "roofline": roofline, "roofline_scan": scan_roofline(phases, hbm_meas),
               "valu_roof": valu, "cpu_baseline": cpu_baseline, "parity_check": parity,
               "vs_cpu_port": round(steps / dt / cpu_baseline["value"], 1), "setup_s": round(t_setup, 1)}
        if skewed:
            out["witness"] = {"aux": inst.num_aux, "boolean_aux": inst.num_boolean,
                              "frac_aux_in_0_1": round(inst.num_boolean / inst.num_aux, 4),
                              "msm_entries": phases["msm_entries"],
                              "note": "SURVEY §8(d) skewed variant: (1 - b) * b = 0 rows (gadgets/src/algebra/boolean.rs:83-90); zero scalars "
                                      "drop out of the digit scan, the ones all land in bucket 1 of the first window"}
        return out
    finally:
        ctx.dev_free(z_dev)
        pk.free()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--log-n", type=int, default=20, help="QAP domain 2^k (k=20 is the BASELINE metric config)")
    ap.add_argument("--curve", default="bn254")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="one blocking zkp_groth16_prove_dev call per step")
    ap.add_argument("--mode", choices=["throughput", "shard"], default="throughput",
                    help="throughput (default, the BASELINE metric): independent proofs per GPU.  shard: ONE proof per step, every "
                         "query base-sharded over the ranks, partial sums all-gathered (RCCL) and folded — BASELINE configs[4]")
    ap.add_argument("--workload", choices=["groth16", "marlin"], default="groth16",
                    help="groth16 (default, the BASELINE metric; its line also carries the other BASELINE configs as blocks "
                         "unless --no-extra-configs / --no-marlin) or marlin: BASELINE configs[3] only")
    ap.add_argument("--no-marlin", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the blocks for BASELINE configs[2] (BLS12-381 2^22), configs[4] on one GPU (BN254 2^24) and the "
                         "skewed-witness variant that the default N=1 BN254 2^20 line carries")
    ap.add_argument("--extra", default="skewed,bls22,bn24", help="which extra blocks to run (comma-separated)")
    ap.add_argument("--budget-s", type=float, default=1400.0,
                    help="wall-clock budget of the whole run: an extra block is skipped (and says so) when the time already "
                         "spent plus its estimate exceeds this")
    ap.add_argument("--devices", default="",
                    help="single-process multi-GPU only: comma-separated device ids for zkp_ctx_create_multi (default 0..gpus-1; "
                         "ids may repeat, e.g. 0,0,0 exercises the 3-rank path on a one-GPU box — not a multi-GPU measurement)")
    ap.add_argument("--single-device-test", action="store_true",
                    help="TEST ONLY: every rank uses cuda:0 and the collectives run over gloo (exercises the N>1 code path "
                         "on a one-GPU box; the number it prints is not a multi-GPU measurement)")
    args = ap.parse_args()
    t_start = time.time()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.devices):
        # launched as ONE process for N GPUs: the in-library multi-GPU path behind the C boundary (zkp_ctx_create_multi)
        return bench_single_process_multi(args)
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    if args.single_device_test:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.single_device_test:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from ckb_zkp_amd import codec, groth16
    from ckb_zkp_amd.api import Context
    from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain
    from ckb_zkp_amd.params import get_curve

    c = get_curve(args.curve)
    ctx = Context(local)
    if args.workload == "marlin":
        m = bench_marlin(ctx, args.curve, reps=max(args.steps if args.steps != 64 else 3, 1), cpu=not args.no_cpu_baseline)
        if rank == 0:
            emit({"metric": "Marlin proofs/sec (2^20 constraints, BN256)", "value": m["value"], "unit": "proofs/s",
                              "n_gpus": 1, "steps": m["runs"], "warmup": 1, "ms_per_step": round(m["s_per_proof"] * 1e3, 2),
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": "u32 limbs (256-bit Montgomery integers)", "data": "synthetic (MiMC-chain R1CS, trapdoor SRS)",
                              "config": {"workload": m["workload"]}, "roofline": m.pop("roofline"), "cpu_baseline": m.pop("cpu_baseline"),
                              "marlin": m})
        return
    t0 = time.time()
    S = samples_for_domain(args.log_n)
    inst = mimc_chain_instance(c, S)
    log(f"instance: MiMC chain S={S}, constraints={inst.num_constraints()}, aux={inst.num_aux} ({time.time()-t0:.1f}s)")
    t0 = time.time()
    params = groth16.generate_parameters(ctx, c, inst, **TOXIC)
    log(f"synthetic proving key from trapdoor (device fixed-base): {time.time()-t0:.1f}s")
    if args.mode == "shard":
        return bench_sharded(args, ctx, c, inst, params, codec.fr_to_mont(inst.z, c).reshape(-1, 4), rank, world, local)
    t0 = time.time()
    pk = groth16.ProvingKey(ctx, params, inst)
    key_upload_s = round(time.time() - t0, 2)
    log(f"key upload + window-table precompute + evaluation-form transforms: {key_upload_s:.1f}s ; domain=2^{pk.domain_size.bit_length()-1}")
    z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
    z_dev = ctx.to_device(z)                                    # inputs resident in HBM before the timed region
    rand_fr = make_rand_fr(c, 1234 + rank)

    def step():
        return pk.prove_raw(z_dev, rand_fr(), rand_fr(), z_on_device=True)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    def blinders(k):
        """the zero-knowledge scalars r, s of k proofs (create_random_proof samples them, prover.rs:110-111): inputs of the
        timed region like the witness, drawn before it"""
        return (np.stack([rand_fr() for _ in range(k)]), np.stack([rand_fr() for _ in range(k)])) if k > 0 else (None, None)

    def steps(k, rs):
        """k proofs through zkp_groth16_prove_batch_dev: the library pipelines consecutive proofs over its lanes
        (every proof is complete — all 3 points back on the host — when the call returns)."""
        if k <= 0:
            return
        if args.no_pipeline:
            for i in range(k):
                pk.prove_raw(z_dev, rs[0][i], rs[1][i], z_on_device=True)
            return
        pk.prove_batch_raw([z_dev] * k, rs[0], rs[1])

    # ---- N > 1: every rank proves the SAME (r, s) once before the timed region; the proofs must agree with rank 0's
    scale_parity = None
    if world > 1:
        import torch.distributed as dist
        fix = make_rand_fr(c, 99)
        p_out, p_inf = pk.prove_raw(z_dev, fix(), fix(), z_on_device=True)
        mine = torch.from_numpy(np.concatenate([p_out.view(np.int64), p_inf.astype(np.int64)]))
        if not args.single_device_test:
            mine = mine.cuda()
        allp = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        scale_parity = {"all_ranks_eq_rank0": bool(all(torch.equal(a, allp[0]) for a in allp)), "ranks": world,
                        "note": "one proof with a fixed (r, s) per rank before the timed region, all-gathered and compared"}

    rs_warm, rs_timed = blinders(args.warmup), blinders(args.steps)
    steps(args.warmup, rs_warm)
    barrier()
    t0 = time.perf_counter()
    steps(args.steps, rs_timed)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.single_device_test else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    primary = args.log_n == 20 and c.name == "bn254"
    solo = rank == 0 and world == 1
    # ---- single-proof latency: blocking zkp_groth16_prove_dev calls, one proof in flight (the latency-oriented stream plan)
    latency = None
    if rank == 0:
        for _ in range(3):
            step()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        latency = {"ms_per_proof": round((time.perf_counter() - t0) / 10 * 1e3, 3),
                   "note": "one proof in flight: zkp_groth16_prove_dev blocking calls (witness resident), host time incl. the read-back"}

    hbm_meas = roofline = phases = roofline_ntt = roofline_scan = valu_roof = with_h2d = None
    if rank == 0:
        try:
            hbm_meas = hbm_peak_block(ctx)
        except Exception as e:
            log("hbm copy peak failed:", repr(e))
        # HBM bytes per launch: PMC counters cannot be read from inside the process; they come from the separate
        # `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this same command recorded under profiles/
        traffic, traffic_src = None, None
        if primary:
            pmc, traffic_src = recorded_latest("pmc_accumulate.json")
            traffic = pmc["traffic_bytes_per_launch"] if pmc else None
        roofline, phases = accumulate_roofline(ctx, pk, c, inst, step, hbm_meas, traffic, traffic_src)
        roofline_ntt = ntt_roofline(ctx, c, args.log_n, hbm_meas)
        roofline_scan = scan_roofline(phases, hbm_meas)
        valu_roof = valu_roof_block(ctx, c, phases, roofline_ntt["valu"]["gmulmod_per_s"])
        # with the witness on the HOST: each proof's assignment crosses PCIe in front of its proof (pinned buffer)
        try:
            zp = torch.from_numpy(z.view(np.int64)).pin_memory()
            k = args.steps
            r = np.stack([rand_fr() for _ in range(k)])
            s_ = np.stack([rand_fr() for _ in range(k)])
            pk.prove_batch_raw([zp.data_ptr()] * 8, r[:8], s_[:8], z_on_device=False)
            ctx.sync()
            t0 = time.perf_counter()
            pk.prove_batch_raw([zp.data_ptr()] * k, r, s_, z_on_device=False)
            t_h2d = time.perf_counter() - t0
            with_h2d = {"value": round(k / t_h2d, 3), "unit": "proofs/s", "ms_per_proof": round(t_h2d / k * 1e3, 3),
                        "h2d_bytes_per_proof": int(z.nbytes),
                        "note": "zkp_groth16_prove_batch: witness in pinned host memory, copied to the device on the proof's "
                                "lane before each proof (overlaps the other lanes' kernels); circuit synthesis itself "
                                "(Rust closures in the reference) stays on the caller's side and is not timed"}
        except Exception as e:                                   # pinned allocation can fail on odd hosts: report, do not die
            with_h2d = {"error": repr(e)}

    # ---- second half of BASELINE.json's metric: one G1 MSM (uniform scalars, the H-query shape) in Mop/s
    msm_g1 = None
    if solo:
        hb = ctx.upload_bases(c, 1, *params.h_query)
        n_msm = hb.n
        sc = np.frombuffer(np.random.default_rng(7).bytes(32 * n_msm), dtype=np.uint64).reshape(-1, 4).copy()
        sc[:, 3] &= (1 << (c.r.bit_length() - 64 * 3 - 1)) - 1           # < r without bias games: top bits cleared
        sc_dev = ctx.to_device(sc)
        import statistics
        for _ in range(5):
            hb.msm_dev(sc_dev, n_msm)
        ts = []
        for _ in range(30):                                    # per-call host wall (result read back each call): median of 30
            t0 = time.perf_counter()
            hb.msm_dev(sc_dev, n_msm)
            ts.append(time.perf_counter() - t0)
        t_msm = statistics.median(ts)
        # throughput of back-to-back MSMs (PC::commit over a list: 4 in flight on the context's MSM streams)
        jobs = [(sc_dev, n_msm, 0)] * 8
        hb.msm_mont_batch_dev(jobs)
        t0 = time.perf_counter()
        hb.msm_mont_batch_dev(jobs)
        t_batch = (time.perf_counter() - t0) / len(jobs)
        msm_g1 = {"n": n_msm, "ms": round(t_msm * 1e3, 3), "mops": round(n_msm / t_msm / 1e6, 1), "best_ms": round(min(ts) * 1e3, 3),
                  "batched_ms": round(t_batch * 1e3, 3), "batched_mops": round(n_msm / t_batch / 1e6, 1),
                  "note": "zkp_msm_g1_dev, canonical scalars resident in HBM, result (Jacobian) back on the host each call, median of 30 calls; "
                          "batched = 8 such MSMs through zkp_msm_g1_mont_batch_dev (4 in flight), per-MSM time"}
        ctx.dev_free(sc_dev)
        hb.free()

    # ---- CPU baseline + parity (rank 0, N=1 only): oracle/cpu port of the reference algorithm
    cpu_baseline = parity_check = None
    if solo and not args.no_cpu_baseline:
        from oracle import cpu_oracle
        import statistics
        cores = cpu_oracle.hardware_threads()
        # probe a 2^16 sample first: a slow host (this container: 8 cores) gets the bounded, scaled sample instead of the full instance
        k = min(16, args.log_n)
        inst_s = mimc_chain_instance(c, samples_for_domain(k))
        params_s = groth16.generate_parameters(ctx, c, inst_s, **TOXIC)
        z_s = codec.fr_to_mont(inst_s.z, c).reshape(-1, 4)
        scale_s = float(1 << (args.log_n - k))
        t0 = time.perf_counter()
        cpu_oracle.groth16_prove(params_s, inst_s, z_s, rand_fr(), rand_fr(), threads=cores)
        t_probe = time.perf_counter() - t0
        if t_probe * scale_s <= 12.0 * (1 << max(args.log_n - 20, 0)):
            cpu_baseline, parity_check = cpu_port_block(pk, params, inst, z, c, cores, 3, f"full 2^{args.log_n}-domain instance", rand_fr)
        else:
            pk_s = groth16.ProvingKey(ctx, params_s, inst_s)
            cpu_baseline, parity_check = cpu_port_block(pk_s, params_s, inst_s, z_s, c, cores, 3, f"2^{k}-domain sample", rand_fr)
            pk_s.free()
            cpu_baseline["value"] = round(cpu_baseline["value"] / scale_s, 6)
            cpu_baseline["s_per_proof"] = round(cpu_baseline["s_per_proof"] * scale_s, 3)
            cpu_baseline["sample"] = f"scaled x{int(scale_s)} linearly from: " + cpu_baseline["sample"]
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            cpu_oracle.groth16_prove(params_s, inst_s, z_s, rand_fr(), rand_fr(), threads=1)
            ts.append(time.perf_counter() - t0)
        t_one = statistics.median(ts)
        cpu_baseline["threads_1"] = {"threads": 1, "proofs_per_s": round(1.0 / (t_one * scale_s), 6),
                                     "s_per_proof": round(t_one * scale_s, 2),
                                     "sample": f"2^{k}-domain MiMC chain, median of 3 runs {[round(x, 2) for x in ts]} s on 1 thread, "
                                               f"scaled x{int(scale_s)} linearly"}

    table_plan = pk.table_plan() if rank == 0 else None
    ctx.dev_free(z_dev)
    marlin = None
    extra = {}
    if solo and primary:
        pk.free()                                              # the primary key's window tables make room for the next workload
        pk = None
        if not args.no_marlin:
            try:
                marlin = bench_marlin(ctx, "bn254", cpu=not args.no_cpu_baseline)
            except Exception as e:                             # never lose the Groth16 line to a secondary workload
                marlin = {"error": repr(e)}
        if not args.no_extra_configs:
            # (name, key, curve, log_n, steps, warmup, skewed, cpu runs, estimated seconds on the GPU box)
            plan = [("skewed", "bn254_2p20_skewed_witness", "bn254", 20, 24, 8, True, 1, 40),
                    ("bls22", "bls12_381_2p22", "bls12_381", 22, 12, 4, False, 1, 150),
                    ("bn24", "bn254_2p24_single_gpu", "bn254", 24, 6, 2, False, 1, 330)]
            want = set(args.extra.split(","))
            for name, key, cv, lg, st_, wu, sk, cr, est in plan:
                if name not in want:
                    continue
                spent = time.time() - t_start
                if spent + est > args.budget_s:
                    extra[key] = {"skipped": f"{spent:.0f} s spent + {est} s estimated > --budget-s {args.budget_s:.0f}"}
                    continue
                try:
                    t0 = time.time()
                    extra[key] = config_block(ctx, cv, lg, st_, wu, hbm_meas, skewed=sk, cpu_runs=cr)
                    extra[key]["block_s"] = round(time.time() - t0, 1)
                    log(f"[{key}] {extra[key]['value']} proofs/s, parity {extra[key]['parity_check']['device_eq_cpu_port']}, "
                        f"{extra[key]['block_s']} s")
                except Exception as e:
                    extra[key] = {"error": repr(e)}
    if rank == 0:
        proofs = args.steps * world
        out = {
            "metric": "Groth16 proofs/sec (2^20 constraints, BN256)" if primary
            else f"Groth16 proofs/sec (2^{args.log_n} domain, {c.name})",
            "value": round(proofs / dt, 4), "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs (256-bit Montgomery integers)",
            "data": "synthetic (MiMC-chain R1CS, PRF witness, proving key generated from a fixed trapdoor)",
            "config": {"workload": f"Groth16 prove, MiMC-chain R1CS, {inst.num_constraints()} constraints "
                                   f"(domain 2^{args.log_n}), {inst.num_aux} aux, {c.name}, G1 x4 + G2 x1 MSM + 4 NTT (7 in the reference: evaluation-form key)",
                       "curve": c.name, "log_domain": args.log_n, "parallelism": f"independent proofs x{world}",
                       "table_plan": table_plan,
                       "key_upload_s": key_upload_s,
                       "key_form": "evaluation form unless ZKP_H_LAGRANGE=0 / ZKP_C_FOLD=0: H query transformed and C folded into the L "
                                   "query once at upload (outside the timed region, like the window tables); 4 instead of 7 transforms per "
                                   "proof, same proof bytes for every assignment (DESIGN.md section 5)",
                       "pipelining": "none" if args.no_pipeline else f"{os.environ.get('ZKP_LANES', '8 (4 above 2^22)')} proofs in flight per GPU (zkp_groth16_prove_batch_dev), GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}"},
            "roofline": roofline, "cpu_baseline": cpu_baseline if world == 1 else recorded_cpu_baseline(c.name, args.log_n),
            "hbm_peak_measured": hbm_meas,
            "roofline_ntt": roofline_ntt, "roofline_scan": roofline_scan, "valu_roof": valu_roof,
            "with_h2d": with_h2d, "latency": latency, "parity_check": parity_check, "scale_parity": scale_parity,
            "exchange": None if world == 1 else {"kind": "none on the data path (independent proofs per rank, replicated key)",
                                                 "control": ("gloo (TEST ONLY)" if args.single_device_test else "rccl") +
                                                            ": barrier + max-over-ranks all_reduce + the scale_parity all_gather",
                                                 "ranks": world},
            "msm_g1": msm_g1, "marlin_config4": marlin, **extra,
            "phases_ms": phases, "bench_wall_s": round(time.time() - t_start, 1),
        }
        out["summary"] = summary_block(out, marlin, extra)     # LAST key: the driver keeps the tail of the line
        emit(out)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
