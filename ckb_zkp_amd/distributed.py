"""Base-sharded multi-GPU MSM / Groth16 (BASELINE.json configs[4]): one process per GPU, `torch.distributed`
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

MSM is a sum over independent (scalar, base) pairs, so each query is split into `world` contiguous index ranges;
every rank keeps only its slice resident (6 GiB of key at 2^24 -> 768 MiB per GPU at 8 GPUs) and runs the normal
single-GPU MSM on it.  The only exchange step is an all-gather of the partial results — EC addition is not an
RCCL reduction op, and what is exchanged is already one point per MSM (never raw buckets): 5 Jacobian points
(4 x 96 B + 192 B for BN254) per proof, i.e. latency-bound on any topology — followed by a local fold.
The NTTs are replicated (every rank computes h; a 2^24 vector is 512 MiB and a distributed four-step NTT would
ship the whole vector over xGMI).

The engine object abstracts the device so that the partition / gather / fold logic is testable on CPU:
    engine.upload(curve, group, xy, inf) -> handle ; engine.msm(handle, scalars_mont) -> jacobian limbs ;
    engine.fold(curve, group, stacked_jacobians) -> jacobian limbs
`GpuEngine` (below) is the product implementation on top of the C ABI.
"""
from __future__ import annotations

import numpy as np

from . import codec
from .params import get_curve


def shard_bounds(n: int, rank: int, world: int):
    """contiguous, balanced: first (n % world) ranks get one extra element."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GpuEngine:
    def __init__(self, ctx):
        self.ctx = ctx

    def upload(self, curve, group, xy, inf):
        return self.ctx.upload_bases(curve, group, xy, inf)

    def msm(self, handle, scalars_mont):
        # Montgomery scalars: into_repr() is fused into the device digit scan
        return handle.vartime_multiscalar_mul(scalars_mont)

    def msm_dev(self, handle, scalars_dev: int, n: int):
        """scalars already resident in HBM (Montgomery): no PCIe traffic in the step"""
        return handle.msm_mont_dev(scalars_dev, n)

    def fold(self, curve, group, stacked):
        return self.ctx.fold(curve, group, stacked)


def all_gather_points(jac: np.ndarray, world: int, device=None) -> np.ndarray:
    """(w,) uint64 -> (world, w) uint64 via torch.distributed.all_gather (RCCL on GPU tensors, gloo on CPU)."""
    if world == 1:
        return jac.reshape(1, -1)
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(jac.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy().view(np.uint64) for o in out])


class ShardedBases:
    """One query (a_query, h_query, SRS powers, ...) split across ranks."""

    def __init__(self, engine, curve, group, xy: np.ndarray, inf, rank: int, world: int, device=None):
        self.engine, self.curve, self.group = engine, get_curve(curve), group
        self.rank, self.world, self.device = rank, world, device
        self.n = xy.shape[0]
        self.lo, self.hi = shard_bounds(self.n, rank, world)
        sl_inf = None if inf is None else np.ascontiguousarray(inf[self.lo:self.hi])
        self.handle = engine.upload(self.curve, group, np.ascontiguousarray(xy[self.lo:self.hi]), sl_inf)

    def partial(self, scalars_mont: np.ndarray) -> np.ndarray:
        n = min(self.n, scalars_mont.shape[0])           # ark min(len) truncation
        lo, hi = min(self.lo, n), min(self.hi, n)
        return self.engine.msm(self.handle, np.ascontiguousarray(scalars_mont[lo:hi]))

    def partial_dev(self, scalars_dev: int, n_scalars: int) -> np.ndarray:
        """the same on a device-resident scalar vector (pointer to element 0, Montgomery, n_scalars elements)"""
        n = min(self.n, n_scalars)
        lo, hi = min(self.lo, n), min(self.hi, n)
        return self.engine.msm_dev(self.handle, scalars_dev + 32 * lo, hi - lo)

    def msm(self, scalars_mont: np.ndarray) -> np.ndarray:
        """Full MSM, identical on every rank: partial -> all-gather -> fold."""
        parts = all_gather_points(self.partial(scalars_mont), self.world, self.device)
        return self.engine.fold(self.curve, self.group, parts.reshape(-1))

    def free(self):
        if self.handle is not None and hasattr(self.handle, "free"):
            self.handle.free()
        self.handle = None


class ShardedGroth16Prover:
    """create_proof (groth16/src/prover.rs:124-211) with every query sharded by index across ranks."""

    def __init__(self, engine, params, inst, rank: int, world: int, device=None, witness_mapper=None):
        c = self.curve = params.curve
        self.engine, self.rank, self.world, self.device = engine, rank, world, device
        self.inst, self.params = inst, params
        self.witness_mapper = witness_mapper              # callable z_mont -> h_mont (replicated NTT pipeline)
        f = c.fq_limbs

        def ext(q, tail, group):
            xy, inf = q
            w = 2 * f * group
            t_xy = np.zeros((4, w), dtype=np.uint64)
            t_inf = np.ones(4, dtype=np.uint8)
            for k, p in enumerate(tail):
                if p is not None:
                    t_xy[k], t_inf[k] = p, 0
            return np.concatenate([xy.reshape(-1, w), t_xy]), np.concatenate([inf, t_inf])

        mk = lambda q, tail, g: ShardedBases(engine, c, g, *ext(q, tail, g), rank, world, device)
        # same folding as groth16.hip: S = z ++ [1, r, s, -rs]
        self.A = mk(params.a_query, [params.alpha_g1, params.delta_g1, None, None], 1)
        self.B1 = mk(params.b_g1_query, [params.beta_g1, None, params.delta_g1, None], 1)
        self.B2 = mk(params.b_g2_query, [params.beta_g2, None, params.delta_g2, None], 2)
        self.L = mk(params.l_query, [None, None, None, params.delta_g1], 1)
        self.H = ShardedBases(engine, c, 1, params.h_query[0], params.h_query[1], rank, world, device)

    def partial_sums(self, z_mont, h_mont, r: int, s: int) -> np.ndarray:
        c = self.curve
        tail = codec.fr_to_mont([1, r, s, (-(r * s)) % c.r], c).reshape(4, 4)
        S = np.concatenate([z_mont, tail])
        ni = self.inst.num_inputs
        return np.concatenate([self.A.partial(S), self.B1.partial(S), self.B2.partial(S), self.H.partial(h_mont),
                               self.L.partial(S[ni:])])

    def fold_sums(self, gathered: np.ndarray) -> np.ndarray:
        """(world, 5 points) -> 5 folded points (Jacobian limbs concatenated A|B1|B2|H|L)."""
        f = self.curve.fq_limbs
        w1, w2 = 3 * f, 6 * f
        offs = [(0, w1, 1), (w1, w1, 1), (2 * w1, w2, 2), (2 * w1 + w2, w1, 1), (3 * w1 + w2, w1, 1)]
        out = []
        for o, w, g in offs:
            out.append(self.engine.fold(self.curve, g, np.ascontiguousarray(gathered[:, o:o + w]).reshape(-1)))
        return np.concatenate(out)

    def prove_sums_dev(self, ctx, pk_m, z_dev: int, r: int, s: int) -> np.ndarray:
        """Device-resident step: z stays in HBM, the witness map runs on the device (replicated), every partial MSM reads its
        slice of S = z ++ [1, r, s, -rs] / h in place; only 5 partial points leave the device before the all-gather."""
        c = self.curve
        nz = self.inst.num_inputs + self.inst.num_aux
        if getattr(self, "_S", None) is None:
            self._S = ctx.dev_alloc((nz + 4) * 32)
            self._h = ctx.dev_alloc(pk_m.domain_size * 32)
            self._ctx_for_free = ctx
        ctx.d2d(self._S, z_dev, nz * 32)
        ctx.h2d(self._S + nz * 32, codec.fr_to_mont([1, r, s, (-(r * s)) % c.r], c).reshape(4, 4))
        pk_m.witness_map_dev(z_dev, self._h)
        ni = self.inst.num_inputs
        # the five partial MSMs of this rank in ONE call (four in flight on the context's MSM streams)
        jobs = []
        for sb, ptr, cnt in ((self.A, self._S, nz + 4), (self.B1, self._S, nz + 4), (self.B2, self._S, nz + 4),
                             (self.H, self._h, pk_m.domain_size), (self.L, self._S + 32 * ni, nz + 4 - ni)):
            n = min(sb.n, cnt)
            lo, hi = min(sb.lo, n), min(sb.hi, n)
            jobs.append((sb.handle, ptr + 32 * lo, hi - lo, 0))
        part = np.concatenate(ctx.msm_mont_multi_dev(jobs))
        return self.fold_sums(all_gather_points(part, self.world, self.device))

    def free(self):
        """release the resident slices and the cached device buffers"""
        for sb in (self.A, self.B1, self.B2, self.L, self.H):
            sb.free()
        if getattr(self, "_S", None) is not None:
            self._ctx_for_free.dev_free(self._S)
            self._ctx_for_free.dev_free(self._h)
            self._S = self._h = None

    def prove_sums(self, z_mont, r: int, s: int) -> np.ndarray:
        h = self.witness_mapper(z_mont)
        part = self.partial_sums(z_mont, h, r, s)
        return self.fold_sums(all_gather_points(part, self.world, self.device))


class DeviceShardedGroth16Prover:
    """One rank of the device-resident base-sharded prover (BASELINE configs[4]): the product path of `--mode shard`.

        partials = zkp_groth16_prove_partials_dev      (witness map + 5 partial MSMs over this rank's 1/world of the key)
        gathered = all_gather_into_tensor(partials)    (RCCL over xGMI; world x 1.25 KiB for BN254)
        proof    = zkp_groth16_fold_assemble_dev       (slot-wise EC sum over the ranks + prover.rs:192-210)

    The partial sums never leave HBM: the send/receive buffers are device tensors handed to the library by pointer.
    transport="gloo" (tests on a one-GPU box / CPU collectives only) stages the 1.25 KiB through the host."""

    def __init__(self, ctx, params, inst, rank: int, world: int, device=None, transport: str = "nccl"):
        from . import groth16
        self.ctx, self.curve, self.rank, self.world, self.transport = ctx, params.curve, rank, world, transport
        self.pk = groth16.ProvingKey(ctx, params, inst, shard=(rank, world))
        self.pb = groth16.partials_bytes(ctx, self.curve)
        self._torch_bufs = None
        self._raw = None
        if world > 1:
            import torch
            dev = device if device is not None else torch.device("cuda", ctx.device)
            self._torch_bufs = (torch.zeros(self.pb, dtype=torch.uint8, device=dev),
                                torch.zeros(self.pb * world, dtype=torch.uint8, device=dev))
            # torch fills them on ITS stream, the library writes them on its own: without this the zero fill of a fresh process
            # (torch's first kernel, late) could land on top of the first proof's partial sums — seen as a wrong FIRST proof in
            # 4 of 10 runs of tests/dist_worker_gpu.py with three ranks on one GPU
            torch.cuda.current_stream(dev).synchronize()
        else:
            self._raw = ctx.dev_alloc(self.pb)

    def prove(self, z_dev: int, r: int, s: int):
        from . import groth16
        if self.world == 1:
            self.pk.partials_dev(z_dev, r, s, self._raw)
            return groth16.fold_assemble_dev(self.ctx, self.curve, self._raw, 1, r, s)
        import torch
        import torch.distributed as dist
        mine, gathered = self._torch_bufs
        self.pk.partials_dev(z_dev, r, s, mine.data_ptr())          # complete on return (library stream synchronised)
        if self.transport == "gloo":
            parts = [torch.empty(self.pb, dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(parts, mine.cpu())
            gathered.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(gathered, mine)
        torch.cuda.current_stream(gathered.device).synchronize()       # the fold runs on the library's stream
        return groth16.fold_assemble_dev(self.ctx, self.curve, gathered.data_ptr(), self.world, r, s)

    def free(self):
        if self.pk is not None:
            self.pk.free()
            self.pk = None
        if self._raw is not None:
            self.ctx.dev_free(self._raw)
            self._raw = None
        self._torch_bufs = None
