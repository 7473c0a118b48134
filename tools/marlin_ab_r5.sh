#!/bin/bash
# Round-5 Marlin scheduling A/B (GPU box): early product-domain FFTs, short first chunk of the chunked MSMs, early beta evaluations.
# best of 10 proofs per setting; two passes over the settings to expose run-to-run noise.
for pass in 1 2; do
for cfg in "0 1 0" "1 1 0" "0 2 0" "0 4 0" "0 1 1" "0 2 1" "1 2 1"; do set -- $cfg
  echo -n "pass $pass EARLY_FFT=$1 CHUNK_FIRST=$2 EARLY_EVAL=$3  "
  ZKP_MARLIN_EARLY_FFT=$1 ZKP_MSM_CHUNK_FIRST=$2 ZKP_MARLIN_EARLY_EVAL=$3 python bench.py --workload marlin --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=l['marlin']; print(m['value'], m['s_per_proof'], m['phase_ms'], m['verified_by_reference_verifier_restatement'])"
done; done
