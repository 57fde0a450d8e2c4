#!/bin/bash
# Round 3: rocprofv3 kernel stats of config 5 (tiny-yolo-xnor 416 batch 128) and config 2 (yolov3-tiny 416 batch 32)
TAG=${1:-r3sx}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/xnor -o bench -- python $R/bench.py --model tiny-yolo-xnor --size 416 --batch 128 --mode fp32 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-extras > $O/xnor_run.log 2>&1
echo "xnor stats exit $?"
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tiny -o bench -- python $R/bench.py --model yolov3-tiny --size 416 --batch 32 --mode fp32 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-extras > $O/tiny_run.log 2>&1
echo "tiny stats exit $?"
cd $R
for d in xnor tiny; do F=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -14 "$F" | cut -c1-200; tail -1 $O/${d}_run.log | cut -c1-160; done
find $O -name "*kernel_trace.csv" -size +20M -delete
