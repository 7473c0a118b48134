#!/bin/bash
# kernel-trace timeline of ONE stand-alone resident-table MSM (blocking zkp_msm_g1): bash tools/trace_msm.sh <outdir> [curve] [group] [log_n] [win_ms]
OUT=$PWD/$1; CURVE=${2:-bn254}; GROUP=${3:-1}; LG=${4:-20}; WIN=${5:-2.3}
ROOT=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $ROOT/tools/msm_one.py $CURVE $GROUP $LG 6 > $OUT/msm.txt 2> $OUT/msm.err
cd $ROOT
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/rocpd_list.py $DB $WIN 0 > $OUT/list.txt 2>&1
find $OUT -name "*.db" -delete
cat $OUT/msm.txt; cat $OUT/list.txt
