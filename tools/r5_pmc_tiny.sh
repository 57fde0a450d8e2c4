#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes of the yolov3-tiny leg (its dominant kernel changed with the tile cost model)
OUT=gpurun_out/${1:-r5r}; mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
export YL_HEAD_CACHE=/tmp/yl_head_cache
C2="--steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-extras --raw-head --nms 0"
A="--model yolov3-tiny --size 416 --batch 32 --mode fp32"
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmcleg_tiny/$C -o pmc -- python $R/bench.py $A $C2 > $R/$OUT/pmcleg_tiny_$C.log 2>&1 )
  echo "pmcleg tiny $C exit $?"
done
python tools/pmc_summary.py $OUT/pmcleg_tiny > $OUT/pmcleg_tiny_summary.txt 2>&1
cat $OUT/pmcleg_tiny_summary.txt | head -40
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_c2 -o s -- python $R/bench.py $A --no-cpu-baseline --no-e2e --no-extras --steps 20 --warmup 3 > $R/$OUT/stats_c2.json 2> $R/$OUT/stats_c2.err )
F=$(find $OUT/stats_c2 -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $OUT/kernel_stats_c2_yolov3_tiny_416_b32_fp32.csv && head -6 "$F" | cut -c1-200
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
