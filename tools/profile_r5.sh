#!/bin/bash
# Round-5 profile batch (GPU box, repo root):  bash tools/profile_r5.sh        (about 25 GPU-minutes)
#   1. tools/profile_r.sh r05: rocprofv3 --kernel-trace --stats of the default Groth16 bench + FETCH_SIZE / WRITE_SIZE PMC passes
#   2. PMC passes over the stand-alone NTT at 2^20 (VALU utilisation, LDS conflicts, occupancy): tools/pmc_ntt.sh
#   3. Marlin: kernel-trace timeline of one proof (tools/trace_marlin.sh) + FETCH_SIZE / WRITE_SIZE of its accumulate kernel
#   4. the bench lines that go to profiles/: the default line (every BASELINE config as a block), the driver's flags, Marlin alone
# then (here, CPU): python tools/collect_r5.py
set -u
R=r05
ROOT=$(pwd)
O=$ROOT/gpurun_out
bash tools/profile_r.sh $R > $O/profile_$R.log 2>&1
bash tools/pmc_ntt.sh gpurun_out/prof_${R}_ntt20 20 > $O/pmc_ntt20_$R.log 2>&1
WIN=66 bash tools/trace_marlin.sh gpurun_out/prof_${R}_marlin > $O/trace_marlin_$R.log 2>&1
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $O/prof_${R}_marlin/pmc_$C -o pmc -- python $ROOT/bench.py --workload marlin --no-cpu-baseline --steps 1 > $O/prof_${R}_marlin/pmc_$C.json 2> $O/prof_${R}_marlin/pmc_$C.err
  DB=$(find $O/prof_${R}_marlin/pmc_$C -name "*.db" | head -1)
  python $ROOT/tools/rocpd_pmc.py $DB accumulate > $O/prof_${R}_marlin/pmc_$C.txt 2>&1
  find $O/prof_${R}_marlin/pmc_$C -name "*.db" -delete
  find $O/prof_${R}_marlin/pmc_$C -name "*.csv" -size +1M -delete
done
cd $ROOT
python bench.py > $O/bench_full.json 2> $O/bench_full.err
python bench.py --steps 20 --warmup 5 --no-marlin --no-extra-configs > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
python bench.py --workload marlin > $O/marlin.json 2> $O/marlin.err
for f in bench_full bench_driver_flags marlin; do echo $f; tail -1 $O/$f.json | tail -c 900; echo; done
