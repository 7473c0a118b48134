#!/usr/bin/env python3
"""Repeats tests/dist_worker_gpu.py (3 ranks on one GPU) under a parent process that holds its own context, with an environment
switch on / off, and counts the runs whose sharded proof differed from the single-GPU proof.
    python tools/repro_shard_race.py <runs> <ENV=VAL> [<ENV=VAL> ...]"""
import os, subprocess, sys
sys.path.insert(0, ".")
from ckb_zkp_amd.api import Context
import numpy as np
ctx = Context(0)
hold = ctx.to_device(np.zeros((1 << 20, 4), dtype=np.uint64))
runs = int(sys.argv[1])
for sw in sys.argv[2:]:
    env = dict(os.environ, PYTHONPATH=os.getcwd(), GPU_MAX_HW_QUEUES="16")
    for kv in sw.split(","):
        k, v = kv.split("=")
        env[k] = v
    bad = 0
    for i in range(runs):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
               "--master-port", str(29800 + i), "tests/dist_worker_gpu.py"]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            bad += 1
            line = [l for l in out.stderr.splitlines() if "AssertionError" in l]
            print("   fail:", line[:1])
    print(sw, "failures", bad, "of", runs, flush=True)
