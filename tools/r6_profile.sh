#!/bin/bash
# round 6: rocprofv3 kernel stats of the shipped tree, one configuration per run (the average launch duration of the dominant kernel
# must agree with the HIP-event figure of the bench line).  Usage: bash tools/r6_profile.sh <tag>
TAG=${1:-r6p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
export YL_HEAD_CACHE=/tmp/yl_head_cache
C1="--no-cpu-baseline --no-e2e --no-extras"
for leg in "c3_yolov3_608_b64_fp32|--mode fp32 --steps 7 --warmup 2" "c4_yolov3_608_b64_int8|--mode int8 --steps 10 --warmup 2"; do
  T=${leg%%|*}; A=${leg#*|}
  timeout 300 python $R/bench.py $A $C1 --steps 1 --warmup 0 > /dev/null 2>&1
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_$T -o s -- python $R/bench.py $A $C1 > $R/$OUT/stats_$T.json 2> $R/$OUT/stats_$T.err )
  echo "stats $T exit $?"
  F=$(find $OUT/stats_$T -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp "$F" $OUT/kernel_stats_$T.csv && head -6 "$F" | cut -c1-220
  tail -1 $OUT/stats_$T.json | cut -c1-300
done
find $OUT -name "*kernel_trace.csv" -delete
