#!/bin/bash
# round 5, GPU call 7: the ping-pong schedule of K1r (tile 63) against the pinned one (61)
TAG=${1:-r5h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_row3.py -m gpu -q --maxfail=10 -k "oracle or bit_identical" > $OUT/pytest_row3.log 2>&1
echo "pytest exit $?"; tail -6 $OUT/pytest_row3.log | cut -c1-300
for rep in 1 2; do
timeout 300 python tools/sweep_conv.py --batch 64 --tiles 61,63 --iters 30 --only 6,9,12,15 2>&1 | grep -E "^\{" | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print(r['shape'], r['M'], r['C'], r['H'], r['kernel'], '%.3f ms' % r['ms'])" | tee -a $OUT/sweep_pp.txt
done
export YL_HEAD_CACHE=/tmp/yl_head_cache
C1="--mode fp32 --no-cpu-baseline --no-e2e --no-extras --steps 10 --warmup 3"
for leg in "t61|--tile 61" "t63|--tile 63" "t61b|--tile 61" "t63b|--tile 63"; do
  T=${leg%%|*}; A=${leg#*|}
  timeout 300 python bench.py $C1 $A > $OUT/bench_$T.json 2> $OUT/bench_$T.err
  echo "bench $T $(tail -1 $OUT/bench_$T.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", d["roofline"]["kernel"], round(d["roofline"]["avg_launch_ms"],4))' 2>&1 | tail -1)"
done
