#!/bin/bash
# kernel-trace timeline of the Marlin prover (last proof): bash tools/trace_marlin.sh <outdir>
OUT=$PWD/$1; shift
ROOT=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $ROOT/bench.py --workload marlin --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/rocpd_gaps.py $DB ${WIN:-85} 100 > $OUT/gaps.txt 2>&1
python tools/rocpd_timeline.py $DB ${WIN:-85} 1.0 > $OUT/timeline.txt 2>&1
find $OUT -name "*.db" -delete
cat $OUT/gaps.txt
