#!/bin/bash
# Round 3: rocprofv3 kernel stats of the default bench line + FETCH_SIZE / WRITE_SIZE passes (own runs, --kernel-trace
# only) of the two side legs (BASELINE configs 2 and 5).  Usage on the GPU box: bash tools/rounds/round_r3_pmc.sh <tag>
TAG=${1:-r3p}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
COMMON="--mode fp32 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-extras --raw-head --nms 0"
cd /tmp
for leg in "yolov3-tiny 416 32 tiny" "tiny-yolo-xnor 416 128 xnor"; do
  set -- $leg
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$4_$C -o pmc -- python $R/bench.py --model $1 --size $2 --batch $3 $COMMON > $O/pmc_$4_$C.log 2>&1
    echo "pmc $4 $C exit $?"
  done
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -o bench -- python $R/bench.py --mode fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extras --raw-head --nms 0 > $O/rocprof_run.log 2>&1
echo "rocprof stats exit $?"
cd $R
python tools/pmc_summary.py $O > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt | head -60
F=$(find $O/rocprof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -12 "$F" | cut -c1-220
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*counter_collection.csv" -size +30M -delete
