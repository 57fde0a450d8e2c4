#!/bin/bash
# A/B builds of the library on the same GPU box: tools/ab/libyolo2hip_<v>.so for v in $VARIANTS
cp yolo2_light_amd/libyolo2hip.so /tmp/keep.so
for round in 1; do
for v in ${VARIANTS:-old new}; do
  cp tools/ab/libyolo2hip_$v.so yolo2_light_amd/libyolo2hip.so
  echo "== $v"
  timeout 200 python tools/sweep_conv.py --batch 64 --tiles ${TILES:-31} --only ${SHAPES:-9,12,15} --iters 5 2>&1 | grep -E "^\{" | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print(r['shape'], r['M'], r['C'], r['H'], r['tile'], '%.3f ms %.1f TF' % (r['ms'], r['tflops']))
"
done
done
cp /tmp/keep.so yolo2_light_amd/libyolo2hip.so
