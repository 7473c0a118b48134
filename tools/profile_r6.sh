#!/bin/bash
# Round-6 profile batch (GPU box, repo root):  bash tools/profile_r6.sh        (about 25 GPU-minutes)
# Same passes as round 5 (tools/profile_r5.sh); since round 6 bench.py prints ONE compact line (< 4 KB) and writes the full record
# to $ZKP_BENCH_DETAIL, so every run here names its own detail file next to its line (<run>.json + <run>.detail.json).
#   1. rocprofv3 --kernel-trace --stats of the default Groth16 bench + FETCH_SIZE / WRITE_SIZE PMC passes (separate runs)
#   2. PMC passes over the stand-alone NTT at 2^20: tools/pmc_ntt.sh
#   3. Marlin: kernel-trace timeline of one proof + FETCH_SIZE / WRITE_SIZE of its accumulate kernel
#   4. the bench lines that go to profiles/: the default line (every BASELINE config as a block), the driver's flags, Marlin alone
# then (here, CPU): python tools/collect_r5.py r06
set -u
R=r06
ROOT=$(pwd)
O=$ROOT/gpurun_out
OUT=$O/prof_$R
mkdir -p $OUT $O/prof_${R}_marlin
export TMPDIR=/tmp
cd /tmp
ZKP_BENCH_DETAIL=$OUT/stats_bench.detail.json rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $ROOT/bench.py --no-cpu-baseline --no-marlin --no-extra-configs > $OUT/stats_bench.json 2> $OUT/stats.err
for C in FETCH_SIZE WRITE_SIZE; do
  ZKP_BENCH_DETAIL=$OUT/pmc_$C.detail.json rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -o pmc -- python $ROOT/bench.py --steps 3 --warmup 1 --no-pipeline --no-cpu-baseline --no-marlin --no-extra-configs > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
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
head -12 $OUT/kernel_stats.txt
bash tools/pmc_ntt.sh gpurun_out/prof_${R}_ntt20 20 > $O/pmc_ntt20_$R.log 2>&1
M=$O/prof_${R}_marlin
cd /tmp
ZKP_BENCH_DETAIL=$M/bench.detail.json rocprofv3 --kernel-trace -d $M/trace -o t -- python $ROOT/bench.py --workload marlin --no-cpu-baseline > $M/bench.json 2> $M/bench.err
cd $ROOT
DB=$(find $M/trace -name "*.db" | head -1)
python tools/rocpd_gaps.py $DB 66 100 > $M/gaps.txt 2>&1
python tools/rocpd_timeline.py $DB 66 1.0 > $M/timeline.txt 2>&1
find $M -name "*.db" -delete
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  ZKP_BENCH_DETAIL=$M/pmc_$C.detail.json rocprofv3 --kernel-trace --pmc $C -d $M/pmc_$C -o pmc -- python $ROOT/bench.py --workload marlin --no-cpu-baseline --steps 1 > $M/pmc_$C.json 2> $M/pmc_$C.err
  DB=$(find $M/pmc_$C -name "*.db" | head -1)
  python $ROOT/tools/rocpd_pmc.py $DB accumulate > $M/pmc_$C.txt 2>&1
  find $M/pmc_$C -name "*.db" -delete
  find $M/pmc_$C -name "*.csv" -size +1M -delete
done
cd $ROOT
ZKP_BENCH_DETAIL=$O/bench_full.detail.json python bench.py > $O/bench_full.json 2> $O/bench_full.err
ZKP_BENCH_DETAIL=$O/bench_driver_flags.detail.json python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
ZKP_BENCH_DETAIL=$O/marlin.detail.json python bench.py --workload marlin > $O/marlin.json 2> $O/marlin.err
for f in bench_full bench_driver_flags marlin; do echo $f; tail -1 $O/$f.json | wc -c; tail -1 $O/$f.json | tail -c 700; echo; done
