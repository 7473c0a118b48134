"""GPU: the single-process multi-GPU boundary (zkp_ctx_create_multi / zkp_groth16_prove_multi /
zkp_groth16_prove_batch_multi) with every rank on cuda:0 (device ids may repeat), and the PRODUCT's
multi-process sharded prover (distributed.DeviceShardedGroth16Prover, what `bench.py --mode shard` runs) as two real
processes on cuda:0 with the collective over gloo.  Replaces one `create_proof` call (groth16/src/prover.rs:124-211)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from ckb_zkp_amd import codec, groth16
from ckb_zkp_amd.api import MultiContext
from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain
from ckb_zkp_amd.params import get_curve

pytestmark = pytest.mark.gpu
TOXIC = dict(alpha=0x1234567890ABCDEF1, beta=0xFEDCBA09876543211, gamma=0x1111111111111111111,
             delta=0x2222222222222222223, tau=0x3333333333333333335)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("curve,k,world", [("bn254", 12, 2), ("bn254", 11, 3), ("bn254", 10, 8), ("bn254", 4, 5),
                                           ("bls12_381", 9, 3)])
def test_prove_multi_sharded_equals_single_gpu(ctx, curve, k, world):
    """ONE proof base-sharded over `world` ranks inside one process (world >= 3: task-split witness map, a / b / c chains
    on ranks 0 / 1 / 2, h slices fetched by every rank) == the single-GPU proof, host witness and device witnesses."""
    c = get_curve(curve)
    inst = mimc_chain_instance(curve, samples_for_domain(k))
    params = groth16.generate_parameters(ctx, curve, inst, **TOXIC)
    pk = groth16.ProvingKey(ctx, params, inst)
    z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
    m = MultiContext([0] * world)
    try:
        mpk = groth16.MultiProvingKey(m, params, inst, groth16.MULTI_SHARD)
        try:
            zds = [m.member(r).to_device(z) for r in range(world)]
            for r_, s_ in ((0xABCDEF0123456789ABCDEF, 0x13579BDF02468ACE), (0, 0)):
                rm, sm = codec.fr_to_mont([r_], c)[0], codec.fr_to_mont([s_], c)[0]
                out1, inf1 = pk.prove_raw(z, rm, sm)
                for rep in range(2):                                # the second call reuses every buffer
                    out2, inf2 = mpk.prove_raw(z, rm, sm)
                    assert np.array_equal(out1, out2) and np.array_equal(inf1, inf2), (r_, s_, rep)
                out3, inf3 = mpk.prove_raw(zds, rm, sm, z_on_device=True)
                assert np.array_equal(out1, out3) and np.array_equal(inf1, inf3)
            for r, d in enumerate(zds):
                m.member(r).dev_free(d)
            # a sharded multi key refuses the batch entry point
            from ckb_zkp_amd._lib import ZkpError
            with pytest.raises(ZkpError):
                mpk.prove_batch_raw([z], np.stack([rm]), np.stack([sm]))
        finally:
            mpk.free()
    finally:
        m.close()
        pk.free()


def test_prove_multi_replicated_witness_map(ctx, monkeypatch):
    """ZKP_MULTI_WM_SPLIT=0 is read once per process, so the replicated-witness-map variant of the 3-rank step runs in a
    child process and must print the same proof as the task-split one."""
    code = r'''
import numpy as np
from ckb_zkp_amd import codec, groth16
from ckb_zkp_amd.api import MultiContext
from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain
TOXIC = dict(alpha=11, beta=13, gamma=17, delta=19, tau=23)
m = MultiContext([0, 0, 0])
inst = mimc_chain_instance("bn254", samples_for_domain(10))
params = groth16.generate_parameters(m, "bn254", inst, **TOXIC)
mpk = groth16.MultiProvingKey(m, params, inst, groth16.MULTI_SHARD)
c = params.curve
z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
out, inf = mpk.prove_raw(z, codec.fr_to_mont([5], c)[0], codec.fr_to_mont([7], c)[0])
print("PROOF", out.tobytes().hex(), inf.tobytes().hex())
'''
    res = {}
    for split in ("0", "1"):
        env = dict(os.environ, ZKP_MULTI_WM_SPLIT=split, PYTHONPATH=ROOT)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        res[split] = [l for l in out.stdout.splitlines() if l.startswith("PROOF")][0]
    assert res["0"] == res["1"]


def test_prove_multi_one_rank_through_rccl(ctx):
    """VERDICT r4 task 5: the RCCL branch of zkp_groth16_prove_multi (dlsym'd ncclCommInitAll / ncclGroupStart / ncclAllGather /
    ncclGroupEnd, csrc/groth16.hip) EXECUTED on the one GPU this box has: ZKP_MULTI_EXCHANGE=rccl with a one-rank multi context
    (a one-rank communicator is legal), the proof equals the plain single-GPU proof and zkp_groth16_multi_info reports
    exchange = rccl, rccl_ranks = 1.  A child process, so that a misbehaving communicator cannot wedge the suite."""
    code = r'''
import numpy as np
from ckb_zkp_amd import codec, groth16
from ckb_zkp_amd.api import MultiContext
from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain
TOXIC = dict(alpha=11, beta=13, gamma=17, delta=19, tau=23)
m = MultiContext([0])
inst = mimc_chain_instance("bn254", samples_for_domain(12))
params = groth16.generate_parameters(m, "bn254", inst, **TOXIC)
c = params.curve
z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
rm, sm = codec.fr_to_mont([0x1234567], c)[0], codec.fr_to_mont([0x7654321], c)[0]
pk = groth16.ProvingKey(m, params, inst)
want = pk.prove_raw(z, rm, sm)
pk.free()
mpk = groth16.MultiProvingKey(m, params, inst, groth16.MULTI_SHARD)
for rep in range(3):
    got = mpk.prove_raw(z, rm, sm)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), rep
zd = m.member(0).to_device(z)
got = mpk.prove_raw([zd], rm, sm, z_on_device=True)
assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
info = mpk.info()
print("INFO", info["exchange"], info["rccl_ranks"], info["devices"])
mpk.free()
m.close()
'''
    env = dict(os.environ, ZKP_MULTI_EXCHANGE="rccl", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "INFO rccl 1 1" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


@pytest.mark.parametrize("hang", ["init", "probe"])
def test_rccl_watchdog_falls_back_to_peer_copies(ctx, hang):
    """VERDICT r5 task 6a: the first multi-rank ncclCommInitAll / collective is where a mis-configured node hangs instead of failing.
    The RCCL bring-up (dlopen + ncclCommInitAll + a probe all-gather on throw-away streams) runs on a helper thread under a deadline
    (zkp_ctx_config.multi_exchange_timeout_ms); ZKP_DEBUG_RCCL_HANG simulates a ncclCommInitAll that never returns ("init") and a
    collective that never completes ("probe").  Either way the call must come back: the proof equals the single-GPU proof, the
    partial sums travelled by peer copies, zkp_groth16_multi_info says so (exchange 2) and stderr names the watchdog.  The
    configuration goes through zkp_ctx_create_multi_ex, not the environment."""
    code = r'''
import time
import numpy as np
from ckb_zkp_amd import codec, groth16
from ckb_zkp_amd.api import MultiContext
from ckb_zkp_amd.circuits import mimc_chain_instance, samples_for_domain
TOXIC = dict(alpha=11, beta=13, gamma=17, delta=19, tau=23)
m = MultiContext([0], dict(multi_exchange="rccl", multi_exchange_timeout_ms=400))
assert m.config()["multi_exchange"] == 1 and m.config()["multi_exchange_timeout_ms"] == 400
inst = mimc_chain_instance("bn254", samples_for_domain(10))
params = groth16.generate_parameters(m, "bn254", inst, **TOXIC)
c = params.curve
z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
rm, sm = codec.fr_to_mont([0x1234567], c)[0], codec.fr_to_mont([0x7654321], c)[0]
pk = groth16.ProvingKey(m, params, inst)
want = pk.prove_raw(z, rm, sm)
pk.free()
mpk = groth16.MultiProvingKey(m, params, inst, groth16.MULTI_SHARD)
t = time.time()
for rep in range(3):
    got = mpk.prove_raw(z, rm, sm)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), rep
print("INFO", mpk.info()["exchange"], "|", round(time.time() - t, 1))
mpk.free()
m.close()
'''
    env = dict(os.environ, ZKP_DEBUG_RCCL_HANG=hang, PYTHONPATH=ROOT)
    env.pop("ZKP_MULTI_EXCHANGE", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("INFO")][0]
    assert "peer (rccl watchdog gave up)" in line, (line, out.stderr[-1500:])
    assert float(line.split("|")[1]) < 30.0                      # deadline 0.4 s per stage: the three proofs return in seconds
    assert "watchdog" in out.stderr and "peer copies" in out.stderr


@pytest.mark.parametrize("world", [1, 2, 3])
def test_prove_batch_multi_replicated_equals_sequential(ctx, world):
    """Throughput mode through the C boundary: n independent proofs dealt round-robin over `world` ranks (one host thread
    per rank driving its lanes) == n blocking single-GPU proofs, host and device witnesses, n not a multiple of world."""
    import random
    curve = "bn254"
    c = get_curve(curve)
    inst = mimc_chain_instance(curve, samples_for_domain(11), seed=0xC0FFEE)
    params = groth16.generate_parameters(ctx, curve, inst, **TOXIC)
    pk = groth16.ProvingKey(ctx, params, inst)
    z = codec.fr_to_mont(inst.z, c).reshape(-1, 4)
    rnd = random.Random(world)
    m = MultiContext([0] * world)
    try:
        mpk = groth16.MultiProvingKey(m, params, inst, groth16.MULTI_REPLICATE)
        try:
            for n in (1, 7, 20):
                rm = codec.fr_to_mont([rnd.randrange(c.r) for _ in range(n)], c)
                sm = codec.fr_to_mont([rnd.randrange(c.r) for _ in range(n)], c)
                outs, infs = mpk.prove_batch_raw([z] * n, rm, sm)
                zds = [m.member(r).to_device(z) for r in range(world)]
                outs_d, infs_d = mpk.prove_batch_raw([zds[i % world] for i in range(n)], rm, sm, z_on_device=True)
                for r, d in enumerate(zds):
                    m.member(r).dev_free(d)
                for i in range(n):
                    o1, i1 = pk.prove_raw(z, rm[i], sm[i])
                    assert np.array_equal(outs[i], o1) and np.array_equal(infs[i], i1), (n, i)
                    assert np.array_equal(outs_d[i], o1) and np.array_equal(infs_d[i], i1), (n, i)
            from ckb_zkp_amd._lib import ZkpError
            with pytest.raises(ZkpError):                       # a replicated key refuses the sharded entry point
                mpk.prove_raw(z, rm[0], sm[0])
        finally:
            mpk.free()
    finally:
        m.close()
        pk.free()


def test_multi_context_argument_checks(ctx):
    from ckb_zkp_amd._lib import ZkpError
    with pytest.raises(ZkpError):
        MultiContext([0, 9999])                                  # no such device: hard error, members already made are released
    m = MultiContext([0, 0])
    try:
        assert m.num_devices == 2
        with pytest.raises(ZkpError):
            m.member(2)
        # an ordinary context is not a multi root
        inst = mimc_chain_instance("bn254", 3)
        params = groth16.generate_parameters(ctx, "bn254", inst, **TOXIC)
        with pytest.raises(ZkpError):
            groth16.MultiProvingKey(ctx, params, inst, groth16.MULTI_SHARD)
    finally:
        m.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_device_sharded_prover_multiprocess_gloo_on_one_gpu(world):
    """The product's sharded prover as REAL processes: `world` ranks launched by torch.distributed.run, every rank on
    cuda:0, DeviceShardedGroth16Prover.prove with its collective branch (all-gather of the device-resident partial sums,
    here staged over gloo) == the single-GPU proof computed by rank 0 with the whole key."""
    env = dict(os.environ, PYTHONPATH=ROOT, GPU_MAX_HW_QUEUES="16")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker_gpu.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    if out.returncode != 0:                                      # keep the whole child output: pytest truncates the assertion repr
        dump = os.path.join(ROOT, "gpurun_out")
        os.makedirs(dump, exist_ok=True)
        with open(os.path.join(dump, f"sharded_gloo_world{world}_failure.txt"), "w") as f:
            f.write(out.stdout + "\n==== stderr ====\n" + out.stderr)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    # (the ranks share one stdout pipe: their lines can land on one line, so count occurrences, not lines)
    assert out.stdout.count("SHARDED_OK rank") == world, out.stdout[-2000:]


def _bench_line(args, torchrun_world=0, timeout=1500):
    env = dict(os.environ, PYTHONPATH=ROOT, GPU_MAX_HW_QUEUES="16")
    if torchrun_world:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={torchrun_world}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    import json
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0]), out.stderr


def test_bench_torchrun_two_ranks_reports_scale_parity_and_exchange():
    """`bench.py --gpus 2` as the driver launches it (torchrun, one process per rank), both ranks on cuda:0 with the test-only gloo
    control plane: throughput mode and the base-sharded mode print scale_parity (checked before the timed region) and exchange."""
    line, _ = _bench_line(["--gpus", "2", "--steps", "4", "--warmup", "2", "--log-n", "12", "--single-device-test",
                           "--no-cpu-baseline", "--no-marlin", "--no-extra-configs"], torchrun_world=2)
    assert line["n_gpus"] == 2 and line["scale_parity"]["all_ranks_eq_rank0"] is True and line["scale_parity"]["ranks"] == 2
    assert line["exchange"]["ranks"] == 2 and "none on the data path" in line["exchange"]["kind"]
    line, _ = _bench_line(["--gpus", "2", "--steps", "3", "--warmup", "1", "--log-n", "12", "--single-device-test", "--mode", "shard"],
                          torchrun_world=2)
    assert line["scaling"] == "strong" and line["scale_parity"]["sharded_eq_single_gpu"] is True
    assert line["exchange"]["ranks"] == 2 and line["exchange"]["bytes_per_rank"] > 0
    # the N > 1 lines carry a per-device roofline (a `None` there would leave the first SCALE run "unmeasured")
    assert line["roofline"]["frac"] > 0 and line["roofline"]["bound"] == "hbm" and "from" in line["roofline"]


def test_bench_single_process_three_ranks_measures_the_witness_split():
    """`bench.py --devices 0,0,0 --mode shard` (zkp_ctx_create_multi + zkp_groth16_prove_multi): scale_parity holds, the exchange
    falls back to peer copies LOUDLY (RCCL refuses duplicate devices) and the witness-map variant is the measured one: both
    variants were timed on the key's own proofs and the faster was kept."""
    line, err = _bench_line(["--devices", "0,0,0", "--mode", "shard", "--steps", "6", "--warmup", "5", "--log-n", "14"])
    assert line["n_gpus"] == 3 and line["scale_parity"]["sharded_eq_single_gpu"] is True
    ex = line["exchange"]
    assert ex["exchange"] == "peer" and ex["rccl_ranks"] == 0 and ex["devices"] == 3
    assert "RCCL all-gather unavailable" in err
    assert ex["witness_map"] in ("replicated", "split over devices 0..2") and ex["ms_replicated"] > 0 and ex["ms_split"] > 0
    assert (ex["witness_map"] == "replicated") == (ex["ms_replicated"] <= ex["ms_split"])
    line, _ = _bench_line(["--devices", "0,0", "--steps", "4", "--warmup", "2", "--log-n", "12"])
    assert line["n_gpus"] == 2 and line["scale_parity"]["all_devices_eq_device0"] is True
    assert line["roofline"]["frac"] > 0 and line["detail"].endswith("bench_detail.json")
