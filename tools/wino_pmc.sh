#!/bin/bash
# PMC passes over single-layer Winograd launches (tools/sweep_conv.py shapes 9,12,15 at tile 30 and the
# direct kernel at tile 20/22 for comparison).  Usage: bash tools/wino_pmc.sh <tag> [tiles]
TAG=${1:-wino}
TILES=${2:-30}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
for C in "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE TCP_TCC_READ_REQ_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc_$N -o pmc -- python $R/tools/sweep_conv.py --batch 64 --tiles $TILES --only ${SHAPES:-12} --iters 1 > $R/$OUT/pmc_$N.log 2>&1 )
  echo "pmc $N exit $?"
done
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
grep -E "wino|pipe" $OUT/pmc_summary.txt | head -80
find $OUT -name "*kernel_trace.csv" -size +5M -delete
