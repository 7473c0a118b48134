#!/bin/bash
# same-box A/B of ONE environment switch: bash tools/ab_env.sh VAR "v0 v1 ..." [passes=2]   (lone 2^20 / 2^22 MSM, Marlin, Groth16 2^20, single-proof latency)
cd "$(dirname "$0")/.."
VAR=$1; VALS=$2; P=${3:-2}
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d.get('summary',{}).get('latency_ms'))"; }
for pass in $(seq 1 $P); do
  for v in $VALS; do
    export $VAR=$v
    s=$(SWEEP_ONLY_DEFAULT=1 python tools/msm_window_sweep.py bn254 20 2>/dev/null | awk '$1=="default"{print $3, $5}')
    m=$(python bench.py --workload marlin --no-cpu-baseline --steps 8 2>/dev/null | val)
    g=$(python bench.py --no-cpu-baseline --no-marlin --no-extra-configs --steps 64 --warmup 10 2>/dev/null | val)
    echo "pass $pass $VAR=$v  lone MSM 2^20 (median ms, Mop/s): $s   marlin (proofs/s ms -): $m   groth16 2^20 (proofs/s ms latency_ms): $g"
  done
done
