#!/bin/bash
# Exact fabric traffic per kernel: L2 (TCC) memory-side requests by size — separate PMC passes (reads by 32 / 64 / 128 B, writes
# by 64 B / other, L2 hits / misses) over a short non-pipelined run.   bash tools/pmc_bytes.sh <outdir>
OUT=$PWD/$1
ROOT=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i -o pmc -- python $ROOT/bench.py --steps 3 --warmup 1 --no-pipeline --no-cpu-baseline --no-marlin --no-extra-configs > $OUT/p$i.json 2> $OUT/p$i.err
done
cd $ROOT
python tools/rocpd_counts.py $(find $OUT -name "*.db") > $OUT/counts.txt 2>&1
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +1M -delete
head -30 $OUT/counts.txt
