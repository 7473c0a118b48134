#!/bin/bash
# Round-6 A/B of the window rule below 2^20 points (msm.hip pick_window_bits; ZKP_MSM_WIDEN=0 = round(log2 n) of rounds 1-5):
# lone G1 MSMs on resident tables and the pipelined Groth16 prover on small circuits, same box, two passes.
cd "$(dirname "$0")/.."
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for pass in 1 2; do
  for w in 0 1; do
    export ZKP_MSM_WIDEN=$w
    for lg in 12 14 16 18; do
      s=$(SWEEP_ONLY_DEFAULT=1 python tools/msm_window_sweep.py bn254 $lg 2>/dev/null | awk '$1=="default"{print $3}')
      g=$(python bench.py --log-n $lg --no-cpu-baseline --no-marlin --no-extra-configs --steps 64 --warmup 10 2>/dev/null | val)
      echo "pass $pass WIDEN=$w 2^$lg: lone MSM median ms $s   groth16 (proofs/s ms_per_proof): $g"
    done
  done
done
