#!/bin/bash
# Same-box comparison of several environment variants: tools/ab_multi.sh <outdir> <rounds> "<env 1>" "<env 2>" ... [-- bench args]
# (each variant = a quoted list of VAR=value words, "-" = none); alternates the variants within every round, prints proofs/s and latency.
OUT=$1; R=$2; shift 2
VARS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VARS+=("$1"); shift; done
[ "$1" = "--" ] && shift
mkdir -p $OUT
for i in $(seq 1 $R); do
  k=0
  for E in "${VARS[@]}"; do
    k=$((k+1))
    if [ "$E" = "-" ]; then EE=""; else EE="$E"; fi
    env $EE python bench.py --no-cpu-baseline --no-marlin --no-extra-configs "$@" > $OUT/v$k.$i.json 2> $OUT/v$k.$i.err
    python - "v$k[$E]" $OUT/v$k.$i.json <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], r["value"], "lat", r.get("latency", {}).get("ms_per_proof"), "acc_ms", r["roofline"].get("avg_launch_ms"), "h2d", r.get("with_h2d", {}).get("value"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
  done
done
