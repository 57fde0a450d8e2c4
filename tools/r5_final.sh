#!/bin/bash
# round 5, last GPU call: the driver's end-of-round sequence (tools/r5_validate.sh) + rocprofv3 kernel stats of the two legs whose first
# layer moved to K1m (config 5: tiny-yolo-xnor 416 b128; config 4: yolov3 608 b64 INT8)
TAG=${1:-r5z}
OUT=gpurun_out/$TAG
bash tools/r5_validate.sh $TAG
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
export YL_HEAD_CACHE=/tmp/yl_head_cache
C1="--no-cpu-baseline --no-e2e --no-extras"
for leg in "c5_tiny_yolo_xnor_416_b128|--model tiny-yolo-xnor --size 416 --batch 128 --mode fp32 --steps 20 --warmup 3" \
           "c4_yolov3_608_b64_int8|--mode int8 --steps 10 --warmup 2"; do
  T=${leg%%|*}; A=${leg#*|}
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_$T -o s -- python $R/bench.py $A $C1 > $R/$OUT/stats_$T.json 2> $R/$OUT/stats_$T.err )
  echo "stats $T exit $?"
  F=$(find $OUT/stats_$T -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp "$F" $OUT/kernel_stats_$T.csv && head -6 "$F" | cut -c1-200
  tail -1 $OUT/stats_$T.json | cut -c1-160
done
find $OUT -name "*kernel_trace.csv" -delete
rm -rf $OUT/stats_c5_tiny_yolo_xnor_416_b128 $OUT/stats_c4_yolov3_608_b64_int8
