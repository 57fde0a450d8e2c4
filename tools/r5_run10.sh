#!/bin/bash
# round 5: K1x with the pinned schedule (tile 55) against the shipped loop (51) -- stand-alone and in the network
OUT=gpurun_out/${1:-r5m}; mkdir -p $OUT
for rep in 1 2; do
timeout 300 python tools/sweep_conv.py --batch 64 --tiles 51,55 --iters 30 --only 1,4,7,8,10,11,13,14 --variant 1086 2>&1 | grep -E "^\{" | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print(r['shape'], r['M'], r['C'], r['size'], r['stride'], r['H'], r['kernel'], '%.3f ms' % r['ms'])" | tee -a $OUT/sweep_x3_pipe.txt
done
