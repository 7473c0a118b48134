#!/bin/bash
# Round-6 A/B: threads per workgroup of the two level-1 sort passes (msm.hip: sort_hist_kernel<NT>, sort_scatter_staged_kernel<NT>).
# The PMC passes (profiles/r06_sort_stage_ab.txt) say all workgroups are resident at once and a wave waits 63 % of its cycles: the kernel
# lasts as long as one workgroup's serial sub-rounds — so spread a tile over more lanes.  ZKP_SORT_NT_HIST / ZKP_SORT_NT_SCATTER = 256 = rounds 3-5.
cd "$(dirname "$0")/.."
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for pass in 1 2; do
  for v in "256 256" "1024 512" "1024 1024"; do
    set -- $v
    export ZKP_SORT_NT_HIST=$1 ZKP_SORT_NT_SCATTER=$2
    s=$(SWEEP_ONLY_DEFAULT=1 python tools/msm_window_sweep.py bn254 20 2>/dev/null | awk '$1=="default"{print $3, $5}')
    s2=$(SWEEP_ONLY_DEFAULT=1 python tools/msm_window_sweep.py bn254 22 2>/dev/null | awk '$1=="default"{print $3, $5}')
    m=$(python bench.py --workload marlin --no-cpu-baseline --steps 8 2>/dev/null | val)
    g=$(python bench.py --no-cpu-baseline --no-marlin --no-extra-configs --steps 64 --warmup 10 2>/dev/null | val)
    echo "pass $pass hist=$1 scatter=$2  lone MSM 2^20 (median ms, Mop/s): $s   2^22: $s2   marlin (proofs/s ms): $m   groth16 2^20 (proofs/s ms): $g"
  done
done
