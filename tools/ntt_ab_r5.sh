#!/bin/bash
# Round-5 NTT experiment (VERDICT r4 task 4): pass plans with up to 10 radix bits per pass on the 1024-element tile.
#   bash tools/ntt_ab_r5.sh > gpurun_out/ntt_ab_r5.txt 2>&1      (GPU box, repo root)
for S in 7 8 9 10; do
  echo "== ZKP_NTT_SMAX=$S  (bit-exactness: tests/test_gpu_ntt.py + the NTT fuzz against oracle/cpu)"
  ZKP_NTT_SMAX=$S python -m pytest tests/test_gpu_ntt.py "tests/test_gpu_fuzz.py::test_ntt_fuzz_against_cpu_port" -m gpu -x -q 2>&1 | tail -2
  for c in bn254 bls12_381; do
    echo "-- $c"
    ZKP_NTT_SMAX=$S python tools/ntt_time.py $c 18,20,21,22,23,24
  done
done
