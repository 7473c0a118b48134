#!/bin/bash
# One PMC pass, one kernel filter: bash tools/pmc_one.sh <outdir> <kernel substring> <counter> [<counter>]
OUT=$PWD/$1; F=$2; shift; shift
ROOT=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc "$@" -d $OUT/p -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --no-marlin --no-extra-configs > $OUT/p.json 2> $OUT/p.err
cd $ROOT
python tools/rocpd_counts.py $(find $OUT -name "*.db") --filter "$F" > $OUT/counts.txt 2>&1
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +1M -delete
cat $OUT/counts.txt
