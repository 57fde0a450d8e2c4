#!/bin/bash
# Round 3, after the conv_xnor changes (count thresholds, 32-filter tiles): FETCH_SIZE / WRITE_SIZE passes of config 5
# only (own runs, --kernel-trace only).  Usage on the GPU box: bash tools/rounds/round_r3_pmc_xnor.sh <tag>
TAG=${1:-r3px}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
COMMON="--mode fp32 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-extras --raw-head --nms 0"
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_xnor_$C -o pmc -- python $R/bench.py --model tiny-yolo-xnor --size 416 --batch 128 $COMMON > $O/pmc_xnor_$C.log 2>&1
  echo "pmc xnor $C exit $?"
done
cd $R
python tools/pmc_summary.py $O > $O/pmc_summary.txt 2>&1
head -40 $O/pmc_summary.txt
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*counter_collection.csv" -size +30M -delete
