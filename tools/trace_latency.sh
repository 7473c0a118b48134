#!/bin/bash
# kernel-trace timeline of ONE blocking Groth16 proof: bash tools/trace_latency.sh <outdir> [curve] [log_n] [win_ms]
OUT=$PWD/$1; CURVE=${2:-bn254}; LG=${3:-20}; WIN=${4:-9.5}
ROOT=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
GAP_S=0.02 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $ROOT/tools/latency_one.py $CURVE $LG 6 > $OUT/latency.txt 2> $OUT/latency.err
cd $ROOT
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/rocpd_list.py $DB $WIN 0 > $OUT/list.txt 2>&1
python tools/rocpd_gaps.py $DB $WIN 50 > $OUT/gaps.txt 2>&1
find $OUT -name "*.db" -delete
cat $OUT/latency.txt; head -5 $OUT/gaps.txt
