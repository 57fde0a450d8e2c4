#!/bin/bash
# Round-2 GPU call D: full parity suite (yolo fusion, softmax tree, INT8 half-depth panels), INT8 tile A/B, default line
TAG=${1:-r2d}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --durations=5 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/summary.txt; tail -12 $OUT/pytest_gpu.log | cut -c1-300
for T in 0 6 7 3 1; do
  timeout 300 python bench.py --mode int8 --i8-tile $T --steps 10 --warmup 2 --layers --no-cpu-baseline --no-e2e --no-extras > $OUT/ab_int8_t$T.json 2> $OUT/ab_int8_t${T}_layers.txt
  echo "int8 tile $T exit $? $(python -c "import json; d=json.loads(open('$OUT/ab_int8_t$T.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)" | tee -a $OUT/summary.txt
done
timeout 300 python bench.py --mode fp32 --steps 10 --warmup 2 --layers --no-cpu-baseline --no-e2e --no-extras > $OUT/bench_fp32.json 2> $OUT/bench_fp32_layers.txt
echo "fp32 exit $? $(python -c "import json; d=json.loads(open('$OUT/bench_fp32.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)" | tee -a $OUT/summary.txt
timeout 300 python bench.py --mode fp32 --no-fuse --steps 10 --warmup 2 --layers --no-cpu-baseline --no-e2e --no-extras > $OUT/bench_fp32_nofuse.json 2> $OUT/bench_fp32_nofuse_layers.txt
echo "fp32 nofuse exit $? $(python -c "import json; d=json.loads(open('$OUT/bench_fp32_nofuse.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)" | tee -a $OUT/summary.txt
du -sh $OUT
