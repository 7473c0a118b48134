"""Worker of tests/test_gpu_multi.py::test_device_sharded_prover_multiprocess_gloo_on_one_gpu: one rank of the product's
device-resident base-sharded Groth16 prover (every rank on cuda:0, collective over gloo)."""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch
import torch.distributed as dist

from ckb_zkp_amd import codec, groth16
from ckb_zkp_amd.api import Context
from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain
from ckb_zkp_amd.distributed import DeviceShardedGroth16Prover

TOXIC = dict(alpha=0x1234567890ABCDEF1, beta=0xFEDCBA09876543211, gamma=0x1111111111111111111,
             delta=0x2222222222222222223, tau=0x3333333333333333335)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    ctx = Context(0)
    for curve, k in (("bn254", 12), ("bls12_381", 9)):
        inst = mimc_chain_instance(curve, samples_for_domain(k))
        params = groth16.generate_parameters(ctx, curve, inst, **TOXIC)      # same trapdoor -> same key on every rank
        c = params.curve
        z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
        zd = ctx.to_device(z)
        prover = DeviceShardedGroth16Prover(ctx, params, inst, rank, world, device=torch.device("cuda", 0), transport="gloo")
        for r_, s_ in ((0xABCDEF0123456789ABCDEF, 0x13579BDF02468ACE), (0, 0)):
            out, inf = prover.prove(zd, r_, s_)
            # every rank folds the same gathered buffer -> every rank holds the proof; compare across ranks and with rank 0's
            # single-GPU proof
            t = torch.from_numpy(np.concatenate([out.view(np.uint8), inf]).copy())
            got = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(got, t)
            assert all(torch.equal(g, got[0]) for g in got)
            if rank == 0:
                pk = groth16.ProvingKey(ctx, params, inst)
                o1, i1 = pk.prove_raw(zd, codec.fr_to_mont([r_], c)[0], codec.fr_to_mont([s_], c)[0], z_on_device=True)
                pk.free()
                assert np.array_equal(out, o1) and np.array_equal(inf, i1), (curve, r_, s_)
        prover.free()
        ctx.dev_free(zd)
        dist.barrier()
    print(f"SHARDED_OK rank {rank}/{world}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
