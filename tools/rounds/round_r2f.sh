#!/bin/bash
# Round-2 GPU call F: parity suite, Winograd epilogue rewrite + unaligned float2 (variant bit 5) A/B, B=8 per-layer table
TAG=${1:-r2f}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --durations=5 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/summary.txt; tail -12 $OUT/pytest_gpu.log | cut -c1-300
for V in 30 62 30 62; do
  timeout 300 python bench.py --mode fp32 --variant $V --steps 10 --warmup 2 --layers --no-cpu-baseline --no-e2e --no-extras > $OUT/ab_fp32_v$V.json 2> $OUT/ab_fp32_v${V}_layers.txt
  echo "fp32 variant $V exit $? $(python -c "import json; d=json.loads(open('$OUT/ab_fp32_v$V.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)" | tee -a $OUT/summary.txt
done
for B in 8 16; do
  timeout 300 python bench.py --mode fp32 --batch $B --steps 20 --warmup 3 --layers --no-cpu-baseline --no-e2e --no-extras > $OUT/bench_fp32_b$B.json 2> $OUT/bench_fp32_b${B}_layers.txt
  echo "fp32 batch $B exit $? $(python -c "import json; d=json.loads(open('$OUT/bench_fp32_b$B.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)" | tee -a $OUT/summary.txt
done
timeout 300 python bench.py --mode int8 --steps 10 --warmup 2 --layers --no-cpu-baseline --no-e2e --no-extras > $OUT/bench_int8.json 2> $OUT/bench_int8_layers.txt
echo "int8 exit $? $(python -c "import json; d=json.loads(open('$OUT/bench_int8.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)" | tee -a $OUT/summary.txt
timeout 900 python bench.py --layers > $OUT/bench.json 2> $OUT/bench_layers.txt
echo "bench exit $?" | tee -a $OUT/summary.txt
tail -1 $OUT/bench.json | cut -c1-300
du -sh $OUT
