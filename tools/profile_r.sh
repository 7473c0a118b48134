#!/bin/bash
# Round profile: kernel stats of the default bench command + separate PMC passes (FETCH_SIZE, WRITE_SIZE) of a
# non-pipelined run.  Usage (on the GPU box, from the repo root): bash tools/profile_r.sh r02
set -u
R=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $ROOT/bench.py --no-cpu-baseline --no-marlin --no-extra-configs > $OUT/stats_bench.json 2> $OUT/stats.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -o pmc -- python $ROOT/bench.py --steps 3 --warmup 1 --no-pipeline --no-cpu-baseline --no-marlin --no-extra-configs > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
cd $ROOT
DB=$(find $OUT/stats -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_stats.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  DB=$(find $OUT/pmc_$C -name "*.db" | head -1)
  python tools/rocpd_pmc.py $DB > $OUT/pmc_$C.txt 2>&1
done
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +2M -delete
ls -la $OUT
head -30 $OUT/kernel_stats.txt
tail -1 $OUT/stats_bench.json | cut -c1-300
