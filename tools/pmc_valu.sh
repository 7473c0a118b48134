#!/bin/bash
# VALU time per kernel: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU / SQ_BUSY_CYCLES of a short non-pipelined run; tools/valu_budget.py
# turns them into "ms of a fully busy vector ALU per proof" (calibrated on the accumulate kernel, which is VALU-bound).
OUT=$PWD/$1
ROOT=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i -o pmc -- python $ROOT/bench.py --steps 3 --warmup 1 --no-pipeline --no-cpu-baseline --no-marlin --no-extra-configs > $OUT/p$i.json 2> $OUT/p$i.err
done
cd $ROOT
python tools/rocpd_counts.py $(find $OUT -name "*.db") > $OUT/counts.txt 2>&1
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +1M -delete
cat $OUT/counts.txt
