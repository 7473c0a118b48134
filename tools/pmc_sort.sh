#!/bin/bash
# PMC passes over the bucket-sort kernels of ONE stand-alone MSM (tools/msm_one.py): bash tools/pmc_sort.sh <outdir> [log_n=20]
OUT=$PWD/$1; LG=${2:-20}
ROOT=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i -o pmc -- python $ROOT/tools/msm_one.py bn254 1 $LG 6 > $OUT/p$i.txt 2> $OUT/p$i.err
done
cd $ROOT
python tools/rocpd_counts.py $(find $OUT -name "*.db") --filter sort_ > $OUT/counts.txt 2>&1
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +1M -delete
cat $OUT/counts.txt
