#!/bin/bash
# rocprofv3 --kernel-trace --stats of the FP32 legs (configs 3 and 2) on the final tree: the per-kernel averages the bench's HIP events must agree with
TAG=${1:-r5y}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
export YL_HEAD_CACHE=/tmp/yl_head_cache
C1="--no-cpu-baseline --no-e2e --no-extras"
for leg in "c3_yolov3_608_b64_fp32|--mode fp32 --steps 7 --warmup 2" \
           "c2_yolov3_tiny_416_b32_fp32|--model yolov3-tiny --size 416 --batch 32 --mode fp32 --steps 20 --warmup 3"; do
  T=${leg%%|*}; A=${leg#*|}
  timeout 300 python $R/bench.py $A $C1 --layers > $OUT/plain_$T.json 2> $OUT/plain_$T.err
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_$T -o s -- python $R/bench.py $A $C1 > $R/$OUT/stats_$T.json 2> $R/$OUT/stats_$T.err )
  echo "stats $T exit $?"
  F=$(find $OUT/stats_$T -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp "$F" $OUT/kernel_stats_$T.csv && head -5 "$F" | cut -c1-220
  python - <<PY
import json
r = json.loads(open("$OUT/plain_$T.json").read().strip().splitlines()[-1])
print("$T plain run: %.1f img/s" % r["value"], r["roofline"].get("kernel"), r["roofline"].get("avg_launch_ms"))
PY
  rm -rf $OUT/stats_$T
done
