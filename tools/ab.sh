#!/bin/bash
# A/B runs on the GPU box: tools/ab.sh <outdir> <variant> [<variant> ...]   (variant "base" = the shipped library,
# otherwise variants/<variant>/libzkp_accel.so built with ZKP_BUILD_TAG).  Per variant: a short parity run (MSM + Groth16
# subsets) and two pipelined bench runs; prints one line per variant.
OUT=$1; shift
mkdir -p $OUT
for V in "$@"; do
  if [ "$V" = base ]; then unset ZKP_ACCEL_LIB; else export ZKP_ACCEL_LIB=$PWD/variants/$V/libzkp_accel.so; fi
  python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py -x -q -m gpu -k "edge or known_dlog or mini or mimc_chain or exceptional or long_buckets" > $OUT/$V.test.log 2>&1
  T=$(tail -1 $OUT/$V.test.log)
  for i in 1 2; do
    python bench.py --no-cpu-baseline --no-marlin ${AB_BENCH_ARGS:-} > $OUT/$V.bench$i.json 2> $OUT/$V.bench$i.err
  done
  python - "$V" "$T" $OUT/$V.bench1.json $OUT/$V.bench2.json <<'PY'
import json, sys
v, t = sys.argv[1], sys.argv[2]
rs = []
for f in sys.argv[3:]:
    try:
        rs.append(json.loads(open(f).read().strip().splitlines()[-1]))
    except Exception as e:
        rs.append(None)
def g(r, *ks):
    for k in ks:
        r = r.get(k) if isinstance(r, dict) else None
    return r
print(f"{v:12s} tests[{t}] proofs/s", [g(r, "value") for r in rs], "lat", [g(r, "latency", "ms_per_proof") for r in rs],
      "acc_ms", [g(r, "roofline", "avg_launch_ms") for r in rs], "ceil", [g(r, "valu_roof", "g1_accumulate", "ceiling") for r in rs],
      "g1frac", [g(r, "valu_roof", "g1_accumulate", "frac") for r in rs], "ntt", [g(r, "roofline_ntt", "ms_per_transform") for r in rs],
      "msm", [g(r, "msm_g1", "ms") for r in rs])
PY
done
