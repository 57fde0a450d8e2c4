#!/bin/bash
# Same-box A/B of FP32 schedule variants on the bench workload (yolov3 608 b64): bash tools/ab_variants.sh <tag> "62 1086" [repeats]
TAG=${1:-ab}; VARS=${2:-"62"}; REP=${3:-2}
O=gpurun_out/$TAG; mkdir -p $O
C1="--no-cpu-baseline --no-e2e --no-extras --mode fp32 --steps 10 --warmup 3 --layers"
for r in $(seq $REP); do
  for v in $VARS; do
    python bench.py $C1 --variant $v 2> $O/layers_${v}_$r.txt | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('variant', $v, 'img/s', d['value'], r.get('kernel'), 'avg launch ms %.4f' % r.get('avg_launch_ms'), 'frac', r.get('frac'))" | tee -a $O/ab.txt
  done
done
