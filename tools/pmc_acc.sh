#!/bin/bash
# Read-side traffic of the accumulate kernels for one library variant: bash tools/pmc_acc.sh <outdir> [variant]
OUT=$PWD/$1
ROOT=$PWD
[ -n "${2:-}" ] && [ "$2" != base ] && export ZKP_ACCEL_LIB=$PWD/variants/$2/libzkp_accel.so
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --no-marlin --no-extra-configs > $OUT/p$i.json 2> $OUT/p$i.err
done
cd $ROOT
python tools/rocpd_counts.py $(find $OUT -name "*.db") --filter accumulate_kernel > $OUT/counts.txt 2>&1
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +1M -delete
cat $OUT/counts.txt
