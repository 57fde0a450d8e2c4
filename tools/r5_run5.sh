#!/bin/bash
# round 5, GPU call 5: K1r schedules (A requested before the input rows; staging pinned between the MFMAs)
TAG=${1:-r5e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_row3.py -m gpu -q --maxfail=10 > $OUT/pytest_row3.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/pytest_row3.log | cut -c1-200
for rep in 1 2; do
timeout 300 python tools/sweep_conv.py --batch 64 --tiles 66,62,61,63,64,67 --iters 30 --only 3,6,9,12,15 2>&1 | grep -E "^\{" | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print(r['shape'], r['M'], r['C'], r['H'], r['kernel'], '%.3f ms' % r['ms'])" | tee -a $OUT/sweep_sched.txt
done
export YL_HEAD_CACHE=/tmp/yl_head_cache
C1="--mode fp32 --no-cpu-baseline --no-e2e --no-extras --steps 10 --warmup 3"
for leg in "t61|--tile 61" "t63|--tile 63" "t62|--tile 62" "t66|--tile 66" "t61b|--tile 61" "t63b|--tile 63" "def|"; do
  T=${leg%%|*}; A=${leg#*|}
  timeout 300 python bench.py $C1 $A > $OUT/bench_$T.json 2> $OUT/bench_$T.err
  echo "bench $T $(tail -1 $OUT/bench_$T.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", d["roofline"]["kernel"], round(d["roofline"]["avg_launch_ms"],4))' 2>&1 | tail -1)"
done
