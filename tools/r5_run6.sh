#!/bin/bash
# round 5, GPU call 6: fine tiles for small grids (K1r 64x32, K1x 64x64) at 8 images; b8 / b16 / b64 steps
TAG=${1:-r5g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_row3.py tests/test_gpu_parity.py -k "row3 or x3" -m gpu -q --maxfail=10 > $OUT/pytest_sel.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/pytest_sel.log | cut -c1-200
timeout 300 python tools/sweep_conv.py --batch 8 --tiles 0,61,64,66,67 --iters 60 --only 9,12,15 2>&1 | grep -E "^\{" | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print('b8', r['shape'], r['M'], r['C'], r['H'], r['tile'], r['kernel'], '%.3f ms' % r['ms'])" | tee $OUT/sweep_row3_b8.txt
timeout 300 python tools/sweep_conv.py --batch 8 --tiles 0,51,52,54 --iters 60 --only 10,11,13,14,16,17 2>&1 | grep -E "^\{" | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print('b8', r['shape'], r['M'], r['C'], r['size'], r['stride'], r['H'], r['tile'], r['kernel'], '%.3f ms' % r['ms'])" | tee $OUT/sweep_x3_b8.txt
export YL_HEAD_CACHE=/tmp/yl_head_cache
C1="--mode fp32 --no-cpu-baseline --no-e2e --no-extras --steps 10 --warmup 3 --layers"
for b in 8 16 64; do
  timeout 300 python bench.py $C1 --batch $b > $OUT/bench_b$b.json 2> $OUT/layers_b$b.txt
  echo "b$b $(tail -1 $OUT/bench_b$b.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s")' 2>&1 | tail -1)"
done
