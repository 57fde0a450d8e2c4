#!/bin/bash
# round 5, GPU call 4: which loads stall K1r (lab builds 128 / 256), remaining tiles, b8 / INT8 per-layer tables, re-run of the tests that failed in call 3
TAG=${1:-r5d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_row3.py "tests/test_gpu_parity.py::test_maxpool_fusion_whole_network_yolov3_tiny" tests/test_gpu_parity.py -k "row3 or maxpool_fusion or x3 or shortcut_fusion" -m gpu -q --maxfail=10 > $OUT/pytest_sel.log 2>&1
echo "pytest exit $?"; tail -8 $OUT/pytest_sel.log | cut -c1-200
ABFILE=conv_f32_row3 TILES=62 SHAPES=9,12,15 ITERS=30 timeout 600 bash tools/ab_builds.sh run "0 128 256" 0 > $OUT/ablation_loads.txt 2>&1
cat $OUT/ablation_loads.txt
timeout 300 python tools/sweep_conv.py --batch 64 --tiles 62,68,69,64,65 --iters 30 --only 3,9,12,15 2>&1 | grep -E "^\{" | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print(r['shape'], r['M'], r['C'], r['H'], r['kernel'], '%.3f ms' % r['ms'])" | tee $OUT/sweep_tiles.txt
timeout 300 python tools/sweep_conv.py --batch 8 --tiles 62,64,68,69 --iters 60 --only 6,9,12,15 2>&1 | grep -E "^\{" | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print('b8', r['shape'], r['M'], r['C'], r['H'], r['kernel'], '%.3f ms' % r['ms'])" | tee $OUT/sweep_tiles_b8.txt
export YL_HEAD_CACHE=/tmp/yl_head_cache
C1="--no-cpu-baseline --no-e2e --no-extras --steps 10 --warmup 3 --layers"
timeout 300 python bench.py --mode fp32 --batch 8 $C1 > $OUT/bench_fp32_b8.json 2> $OUT/layers_fp32_b8.txt; echo "b8 exit $?"
timeout 300 python bench.py --mode fp32 $C1 > $OUT/bench_fp32_b64.json 2> $OUT/layers_fp32_b64.txt; echo "b64 exit $?"
timeout 300 python bench.py --mode int8 $C1 > $OUT/bench_int8_b64.json 2> $OUT/layers_int8_b64.txt; echo "int8 exit $?"
for f in fp32_b8 fp32_b64 int8_b64; do tail -1 $OUT/bench_$f.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', round(d['value'],1), 'img/s')"; done
