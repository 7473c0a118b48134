#!/bin/bash
# Kernel-trace timeline of the pipelined bench + union / accumulate coverage (tools/rocpd_overlap.py), and the per-kernel stats.
#   bash tools/trace_overlap.sh <outdir> [bench args]
OUT=$PWD/$1; shift
ROOT=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $ROOT/bench.py --no-cpu-baseline --no-marlin --steps 48 --warmup 8 "$@" > $OUT/trace_bench.json 2> $OUT/trace.err
cd $ROOT
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/rocpd_overlap.py $DB > $OUT/overlap.txt 2>&1
python tools/rocpd_stats.py $DB > $OUT/kernel_stats.txt 2>&1
python tools/rocpd_fill.py $DB > $OUT/fill.txt 2>&1
[ -n "${KEEP_DB:-}" ] && gzip -c $DB > $OUT/trace.db.gz
find $OUT -name "*.db" -delete
cat $OUT/overlap.txt $OUT/fill.txt
head -40 $OUT/kernel_stats.txt
