#!/bin/bash
# per-kernel stats of a bench command: bash tools/kstats.sh <outdir> <bench args...>
OUT=$PWD/$1; shift
ROOT=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $ROOT/bench.py --no-cpu-baseline --no-marlin "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_stats.txt 2>&1
find $OUT -name "*.db" -delete
head -45 $OUT/kernel_stats.txt | cut -c1-60,88-150
