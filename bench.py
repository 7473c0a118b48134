#!/usr/bin/env python3
"""Benchmark: Groth16 proofs/sec on a 2^20-constraint MiMC-chain R1CS over BN254 (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: one complete Groth16 proof
(sparse A/B/C·z, 7 NTTs + pointwise = witness_map, five MSMs incl. the G2 one, assembly -> 3 affine points)
with the proving key, the circuit matrices and the witness z already resident in HBM.  Multi-GPU: one process
per GPU, every rank proves independent proofs with the same resident key (weak scaling, no data-path
collective); `value` = proofs of all ranks / max-over-ranks time.

The JSON line also carries
  roofline      the dominant kernel (MSM bucket accumulation): algorithmic HBM bytes per launch / HIP-event time
  cpu_baseline  the C++ restatement of the reference algorithm (oracle/cpu, "port") on the host cores,
                on a bounded smaller instance, scaled linearly to the 2^20 instance (rank 0, N=1 only).
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
        print(json.dumps({
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
            "roofline": None, "cpu_baseline": None}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--log-n", type=int, default=20, help="QAP domain 2^k (k=20 is the BASELINE metric config)")
    ap.add_argument("--curve", default="bn254")
    ap.add_argument("--cpu-log-n", type=int, default=16, help="size of the bounded cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="one blocking zkp_groth16_prove_dev call per step")
    ap.add_argument("--mode", choices=["throughput", "shard"], default="throughput",
                    help="throughput (default, the BASELINE metric): independent proofs per GPU.  shard: ONE proof per step, every "
                         "query base-sharded over the ranks, partial sums all-gathered (RCCL) and folded — BASELINE configs[4]")
    ap.add_argument("--single-device-test", action="store_true",
                    help="TEST ONLY: every rank uses cuda:0 and the collectives run over gloo (exercises the N>1 code path "
                         "on a one-GPU box; the number it prints is not a multi-GPU measurement)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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
    log(f"key upload + window-table precompute: {time.time()-t0:.1f}s ; domain=2^{pk.domain_size.bit_length()-1}")
    z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
    z_dev = ctx.to_device(z)                                    # inputs resident in HBM before the timed region
    rng = np.random.default_rng(1234 + rank)

    def rand_fr():
        v = int.from_bytes(rng.bytes(32), "little") % c.r
        return codec.fr_to_mont([v], c)[0]

    def step():
        return pk.prove_raw(z_dev, rand_fr(), rand_fr(), z_on_device=True)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    def steps(k):
        """k proofs through zkp_groth16_prove_batch_dev: the library pipelines consecutive proofs over two lanes
        (every proof is complete — all 3 points back on the host — when the call returns)."""
        if k <= 0:
            return
        if args.no_pipeline:
            for _ in range(k):
                step()
            return
        r = np.stack([rand_fr() for _ in range(k)])
        s = np.stack([rand_fr() for _ in range(k)])
        pk.prove_batch_raw([z_dev] * k, r, s)

    steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.single_device_test else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- roofline of the dominant kernel (bucket accumulation), measured live with HIP events on the ctx stream
    roofline = None
    phases = None
    if rank == 0:
        ctx.set_profiling(True)
        acc_ms, launches, tm_last = 0.0, 0, None
        for _ in range(2):
            step()
            tm_last = pk.last_timing()
            acc_ms += tm_last["ms_msm_accumulate"]
            launches += tm_last["msm_accumulate_launches"]
        ctx.set_profiling(False)
        phases = tm_last
        fq = c.fq_limbs * 8
        nz = inst.num_inputs + inst.num_aux
        # algorithmic bytes: scalars read once (32 B) + affine bases read once, per MSM
        msm_bytes = [(nz + 4) * (32 + 2 * fq), (nz + 4) * (32 + 2 * fq), (nz + 4) * (32 + 4 * fq),
                     (pk.domain_size - 1) * (32 + 2 * fq), (inst.num_aux + 4) * (32 + 2 * fq)]
        bytes_per_launch = sum(msm_bytes) / 5.0
        avg_ms = acc_ms / max(launches, 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None            # HBM bytes per launch from the PMC passes recorded under profiles/ (not live)
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_accumulate.json")
        if args.log_n == 20 and c.name == "bn254" and os.path.exists(pmc):
            traffic = json.load(open(pmc))["traffic_bytes_per_launch"]
        roofline = {"bound": "hbm", "kernel": "accumulate_kernel (MSM bucket accumulation, 5 launches/proof)",
                    "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 5),
                    "traffic": traffic, "avg_launch_ms": round(avg_ms, 4),
                    "algorithmic_bytes_per_launch": int(bytes_per_launch),
                    "note": "integer-VALU bound (DESIGN.md); traffic = FETCH_SIZE*2 + WRITE_SIZE per launch (rocprofv3 PMC, profiles/): "
                            "the window-table design gathers each base W=13 times at 128-B fabric granularity"}

    # ---- second half of BASELINE.json's metric: one G1 MSM (uniform scalars, the H-query shape) in Mop/s
    msm_g1 = None
    if rank == 0 and world == 1:
        hb = ctx.upload_bases(c, 1, *params.h_query)
        n_msm = hb.n
        sc = np.frombuffer(np.random.default_rng(7).bytes(32 * n_msm), dtype=np.uint64).reshape(-1, 4).copy()
        sc[:, 3] &= (1 << (c.r.bit_length() - 64 * 3 - 1)) - 1           # < r without bias games: top bits cleared
        sc_dev = ctx.to_device(sc)
        for _ in range(2):
            hb.msm_dev(sc_dev, n_msm)
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            hb.msm_dev(sc_dev, n_msm)
        t_msm = (time.perf_counter() - t0) / reps
        # throughput of back-to-back MSMs (PC::commit over a list: 4 in flight on the context's MSM streams)
        jobs = [(sc_dev, n_msm, 0)] * 8
        hb.msm_mont_batch_dev(jobs)
        t0 = time.perf_counter()
        hb.msm_mont_batch_dev(jobs)
        t_batch = (time.perf_counter() - t0) / len(jobs)
        msm_g1 = {"n": n_msm, "ms": round(t_msm * 1e3, 3), "mops": round(n_msm / t_msm / 1e6, 1),
                  "batched_ms": round(t_batch * 1e3, 3), "batched_mops": round(n_msm / t_batch / 1e6, 1),
                  "note": "zkp_msm_g1_dev, canonical scalars resident in HBM, result (Jacobian) back on the host each call; "
                          "batched = 8 such MSMs through zkp_msm_g1_mont_batch_dev (4 in flight), per-MSM time"}
        ctx.dev_free(sc_dev)
        hb.free()

    # ---- CPU baseline (rank 0, N=1 only): oracle/cpu port of the reference algorithm on a bounded sample
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu_oracle
        cores = cpu_oracle.hardware_threads()
        k = min(args.cpu_log_n, args.log_n)
        inst_s = mimc_chain_instance(c, samples_for_domain(k))
        params_s = groth16.generate_parameters(ctx, c, inst_s, **TOXIC)
        z_s = codec.fr_to_mont(inst_s.z, c).reshape(-1, 4)
        t0 = time.perf_counter()
        _, _, ph = cpu_oracle.groth16_prove(params_s, inst_s, z_s, rand_fr(), rand_fr(), threads=cores)
        t_cpu = time.perf_counter() - t0
        scale = float(1 << (args.log_n - k))
        sample = (f"one Groth16 proof of the 2^{k}-domain MiMC chain ({inst_s.num_constraints()} constraints) in "
                  f"{t_cpu:.2f}s on {cores} threads; scaled x{int(scale)} linearly to 2^{args.log_n}")
        if k < args.log_n and t_cpu * scale <= 40.0:
            # cheap enough on this host: time the FULL instance instead of extrapolating
            t0 = time.perf_counter()
            _, _, ph = cpu_oracle.groth16_prove(params, inst, z, rand_fr(), rand_fr(), threads=cores)
            t_cpu, scale = time.perf_counter() - t0, 1.0
            sample = (f"one Groth16 proof of the full 2^{args.log_n}-domain instance ({inst.num_constraints()} "
                      f"constraints) in {t_cpu:.2f}s on {cores} threads (no extrapolation)")
        cpu_baseline = {"value": round(1.0 / (t_cpu * scale), 6), "unit": "proofs/s", "cores": cores, "kind": "port",
                        "sample": sample + "; oracle/cpu = C++ restatement of ark-ec/ark-poly 0.2 (window-parallel "
                                           "Pippenger with the arkworks window rule, radix-2 NTT), not the Rust binary",
                        "phase_ms": [round(x, 1) for x in ph.tolist()]}

    if rank == 0:
        proofs = args.steps * world
        out = {
            "metric": "Groth16 proofs/sec (2^20 constraints, BN256)" if (args.log_n == 20 and c.name == "bn254")
            else f"Groth16 proofs/sec (2^{args.log_n} domain, {c.name})",
            "value": round(proofs / dt, 4), "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs (256-bit Montgomery integers)",
            "data": "synthetic (MiMC-chain R1CS, PRF witness, proving key generated from a fixed trapdoor)",
            "config": {"workload": f"Groth16 prove, MiMC-chain R1CS, {inst.num_constraints()} constraints "
                                   f"(domain 2^{args.log_n}), {inst.num_aux} aux, {c.name}, G1 x4 + G2 x1 MSM + 7 NTT",
                       "curve": c.name, "log_domain": args.log_n, "parallelism": f"independent proofs x{world}",
                       "pipelining": "none" if args.no_pipeline else f"{os.environ.get('ZKP_LANES', '8 (4 above 2^22)')} proofs in flight per GPU (zkp_groth16_prove_batch_dev), GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}"},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "msm_g1": msm_g1, "phases_ms": phases,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
