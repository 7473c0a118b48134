#!/bin/bash
# Round-3 profile batch (GPU box, repo root): kernel stats + FETCH/WRITE PMC passes (tools/profile_r.sh), request-size PMC passes,
# timeline overlap, and the bench lines that go to profiles/.   bash tools/profile_r3.sh
set -u
R=r03
bash tools/profile_r.sh $R > gpurun_out/profile_$R.log 2>&1
bash tools/pmc_bytes.sh gpurun_out/prof_${R}_bytes > gpurun_out/pmc_bytes_$R.log 2>&1
bash tools/trace_overlap.sh gpurun_out/prof_${R}_overlap > gpurun_out/overlap_$R.log 2>&1
python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
python bench.py --steps 20 --warmup 5 --no-marlin > gpurun_out/bench_driver_flags.json 2> gpurun_out/bench_driver_flags.err
python bench.py --curve bls12_381 --log-n 22 --steps 12 --warmup 4 --no-marlin > gpurun_out/bls22.json 2> gpurun_out/bls22.err
python bench.py --workload marlin > gpurun_out/marlin.json 2> gpurun_out/marlin.err
python bench.py --log-n 24 --steps 6 --warmup 2 --no-cpu-baseline --no-marlin > gpurun_out/bn24.json 2> gpurun_out/bn24.err
ZKP_TABLE_BUDGET_GB=24 python bench.py --log-n 24 --steps 6 --warmup 2 --no-cpu-baseline --no-marlin > gpurun_out/bn24_budget24.json 2> gpurun_out/bn24_budget24.err
for f in bench_full bench_driver_flags bls22 marlin bn24 bn24_budget24; do echo $f; tail -1 gpurun_out/$f.json | cut -c1-220; done
