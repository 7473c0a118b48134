mkdir -p gpurun_out/r4k
for v in "base" "ZKP_G1_ACC_WAVES=3" "ZKP_G2_ACC_OCC=2" "ZKP_G1_ACC_WAVES=3 ZKP_G2_ACC_OCC=2"; do
  tag=$(echo "$v" | tr ' =' '__')
  if [ "$v" = base ]; then env python bench.py --curve bls12_381 --log-n 22 --steps 12 --warmup 4 --no-marlin --no-extra-configs --no-cpu-baseline > gpurun_out/r4k/bls_$tag.json 2>/dev/null
  else env $v python bench.py --curve bls12_381 --log-n 22 --steps 12 --warmup 4 --no-marlin --no-extra-configs --no-cpu-baseline > gpurun_out/r4k/bls_$tag.json 2>/dev/null; fi
  python -c "
import json,sys;d=json.loads(open('gpurun_out/r4k/bls_$tag.json').read().strip().split('\n')[-1]);print('$v', d['value'], d['ms_per_step'], d['latency']['ms_per_proof'], d['valu_roof']['g1_accumulate'], d['valu_roof']['g2_accumulate']['frac'], d['phases_ms']['ms_msm_acc'])"
done
