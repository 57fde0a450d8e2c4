#!/bin/bash
OUT=gpurun_out/${1:-r5q}; mkdir -p $OUT
export YL_HEAD_CACHE=/tmp/yl_head_cache
C1="--mode fp32 --no-cpu-baseline --no-e2e --no-extras --steps 20 --warmup 3 --layers"
timeout 300 python bench.py $C1 --model yolov3-tiny --size 416 --batch 32 > $OUT/bench_tiny.json 2> $OUT/layers_tiny.txt; tail -1 $OUT/bench_tiny.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("tiny", round(d["value"],1), {k:(v["launches"],round(v["ms_per_step"],3)) for k,v in r["by_kernel"].items()})'
for b in 8 16 32 64; do
  timeout 300 python bench.py --mode fp32 --no-cpu-baseline --no-e2e --no-extras --steps 10 --warmup 3 --batch $b > $OUT/bench_b$b.json 2> $OUT/err_b$b.txt
  echo "b$b $(tail -1 $OUT/bench_b$b.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["value"],1), {k:(v["launches"],round(v["ms_per_step"],3)) for k,v in r["by_kernel"].items() if "row3" in k})')"
done
