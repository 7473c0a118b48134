#!/bin/bash
# A/B of an environment switch on the GPU box: tools/ab_env.sh <outdir> <rounds> "<env A>" "<env B>" [bench args...]
# (e.g. tools/ab_env.sh gpurun_out/ab 3 "ZKP_CHAIN_LH=0" "ZKP_CHAIN_LH=1"); alternates A, B, A, B ... and prints proofs/s per run.
OUT=$1; R=$2; A=$3; B=$4; shift 4
mkdir -p $OUT
for i in $(seq 1 $R); do
  for V in A B; do
    if [ $V = A ]; then E=$A; else E=$B; fi
    env $E python bench.py --no-cpu-baseline --no-marlin "$@" > $OUT/$V.$i.json 2> $OUT/$V.$i.err
    python - "$V[$E]" $OUT/$V.$i.json <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], r["value"], "lat", r.get("latency", {}).get("ms_per_proof"), "acc_ms", r["roofline"].get("avg_launch_ms"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
  done
done
