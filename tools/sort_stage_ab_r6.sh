#!/bin/bash
# Round-6 A/B (VERDICT r5 task 3a): level-1 scatter at more workgroups per CU.  The staged scatter's LDS stage (SORT_STAGE_BYTES,
# msm.hip) is what limits sort_scatter_staged_kernel to 2 workgroups (8 waves) per CU; variants/stage40 and variants/stage28 are
# builds with a 40 KiB / 28 KiB stage (3 / 5 workgroups per CU, sub-rounds of 256 instead of 512 scalars):
#   ZKP_BUILD_TAG=stage28 ZKP_BUILD_DEFS_msm="-DZKP_SORT_STAGE_BYTES=28672" python -m ckb_zkp_amd.build
# Marlin config 4 (lone 6.3 M-point MSMs: the sorts are exposed) and the pipelined Groth16 line, two passes, same box.
cd "$(dirname "$0")/.."
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for pass in 1 2; do
  for v in default stage40 stage28; do
    if [ $v = default ]; then unset ZKP_ACCEL_LIB; else export ZKP_ACCEL_LIB=$PWD/variants/$v/libzkp_accel.so; fi
    m=$(python bench.py --workload marlin --no-cpu-baseline --steps 8 2>/dev/null | val)
    g=$(python bench.py --no-cpu-baseline --no-marlin --no-extra-configs --steps 64 --warmup 10 2>/dev/null | val)
    s=$(SWEEP_ONLY_DEFAULT=1 python tools/msm_window_sweep.py bn254 20 2>/dev/null | awk '$1=="default"{print $3, $5}')
    echo "pass $pass $v  marlin(proofs/s ms): $m   groth16 2^20 (proofs/s ms): $g   lone MSM 2^20 (median ms, Mop/s): $s"
  done
done
