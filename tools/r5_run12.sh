#!/bin/bash
OUT=gpurun_out/${1:-r5o}; mkdir -p $OUT
for rep in 1 2; do
timeout 300 python tools/sweep_conv.py --set yolov3-tiny-416 --batch 32 --tiles 0,61,62,64,65,67,69 --iters 100 --only 0,1,3 2>&1 | grep -E "^\{" | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print('tiny b32', r['shape'], r['M'], r['C'], r['H'], r['tile'], r['kernel'], '%.3f ms' % r['ms'])" | tee -a $OUT/sweep_tiny_row3.txt
done
timeout 300 python tools/sweep_conv.py --set yolov3-tiny-416 --batch 32 --tiles 0,51,52,54 --iters 100 --only 4,5,6,7 2>&1 | grep -E "^\{" | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print('tiny b32', r['shape'], r['M'], r['C'], r['H'], r['tile'], r['kernel'], '%.3f ms' % r['ms'])" | tee -a $OUT/sweep_tiny_x3.txt
