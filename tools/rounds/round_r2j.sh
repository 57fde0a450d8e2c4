#!/bin/bash
# Round-2 GPU call J (last 2 GPU-minutes): rocprofv3 kernel stats of the FP32 leg of the final build
OUT=gpurun_out/${1:-r2j}; mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
( cd /tmp && timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/rocprof -o bench -- python $R/bench.py --mode fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extras --raw-head --nms 0 > $R/$OUT/rocprof_run.log 2>&1 )
echo "rocprof exit $?"
find $OUT -name "*kernel_trace.csv" -delete
head -4 $OUT/rocprof/bench_kernel_stats.csv | cut -c1-200
