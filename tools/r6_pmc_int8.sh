#!/bin/bash
# round 6: wave-state / LDS / cache counters of the INT8 convolution (conv_i8_mfma.hip) in the network, one step of yolov3-608 batch 64
# -quantized.  Usage: bash tools/r6_pmc_int8.sh <tag>
TAG=${1:-r6pmci8}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
export YL_HEAD_CACHE=/tmp/yl_head_cache
A="--mode int8 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-extras --raw-head --nms 0"
timeout 300 python $R/bench.py $A > /dev/null 2>&1
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" \
         "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/p$i -o pmc -- python $R/bench.py $A > $R/$OUT/p$i.log 2>&1 )
  echo "pass $i exit $?"
done
python $R/tools/pmc_dispatch.py $OUT "conv_i8_mfma_kernel<128, 128" | cut -c1-700 > $OUT/i8_pmc.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +2M -delete
wc -l $OUT/i8_pmc.txt
