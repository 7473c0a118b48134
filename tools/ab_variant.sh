#!/bin/bash
# same-box A/B of the in-tree library against variants/<tag>/libzkp_accel.so: lone 2^20 / 2^22 MSM, Marlin config 4, pipelined Groth16 2^20
#   bash tools/ab_variant.sh <tag> [passes=2]
cd "$(dirname "$0")/.."
TAG=$1; P=${2:-2}
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for pass in $(seq 1 $P); do
  for v in $TAG default; do
    if [ $v = default ]; then unset ZKP_ACCEL_LIB; else export ZKP_ACCEL_LIB=$PWD/variants/$v/libzkp_accel.so; fi
    s=$(SWEEP_ONLY_DEFAULT=1 python tools/msm_window_sweep.py bn254 20 2>/dev/null | awk '$1=="default"{print $3, $5}')
    s2=$(SWEEP_ONLY_DEFAULT=1 python tools/msm_window_sweep.py bn254 22 2>/dev/null | awk '$1=="default"{print $3, $5}')
    m=$(python bench.py --workload marlin --no-cpu-baseline --steps 8 2>/dev/null | val)
    g=$(python bench.py --no-cpu-baseline --no-marlin --no-extra-configs --steps 64 --warmup 10 2>/dev/null | val)
    echo "pass $pass $v  lone MSM 2^20 (median ms, Mop/s): $s   2^22: $s2   marlin (proofs/s ms): $m   groth16 2^20 (proofs/s ms): $g"
  done
done
