#!/bin/bash
# Round-2 GPU call B: BF16 / XNOR-fusion tests, Winograd-from-32-channels A/B, BF16 bench leg
TAG=${1:-r2b}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_int8_xnor.py -m gpu -q --maxfail=10 -k "bf16 or shortcut_fusion" > $OUT/pytest_new.log 2>&1
echo "pytest new exit $?" | tee -a $OUT/summary.txt; tail -25 $OUT/pytest_new.log
for V in 14 30 14 30; do
  timeout 300 python bench.py --mode fp32 --variant $V --steps 10 --warmup 2 --layers --no-cpu-baseline --no-e2e --no-extras > $OUT/ab_fp32_v$V.json 2> $OUT/ab_fp32_v${V}_layers.txt
  echo "fp32 variant $V exit $? $(python -c "import json; d=json.loads(open('$OUT/ab_fp32_v$V.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)" | tee -a $OUT/summary.txt
done
timeout 300 python bench.py --mode bf16 --steps 10 --warmup 2 --layers --no-cpu-baseline --no-e2e --no-extras > $OUT/bench_bf16.json 2> $OUT/bench_bf16_layers.txt
echo "bf16 exit $? $(python -c "import json; d=json.loads(open('$OUT/bench_bf16.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)" | tee -a $OUT/summary.txt
timeout 300 python bench.py --mode int8 --steps 10 --warmup 2 --layers --no-cpu-baseline --no-e2e --no-extras > $OUT/bench_int8.json 2> $OUT/bench_int8_layers.txt
echo "int8 exit $? $(python -c "import json; d=json.loads(open('$OUT/bench_int8.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)" | tee -a $OUT/summary.txt
timeout 300 python bench.py --model yolov3-tiny --size 416 --batch 32 --mode fp32 --steps 20 --warmup 3 --layers --no-cpu-baseline --no-e2e --no-extras > $OUT/bench_tiny.json 2> $OUT/bench_tiny_layers.txt
echo "tiny exit $? $(python -c "import json; d=json.loads(open('$OUT/bench_tiny.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)" | tee -a $OUT/summary.txt
du -sh $OUT
