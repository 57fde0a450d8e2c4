#!/bin/bash
# Round-2 GPU call G: parity suite, branch-free Winograd epilogue + carry-decode first-layer kernel, K1 tile 13 A/B
TAG=${1:-r2g}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --durations=5 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/summary.txt; tail -12 $OUT/pytest_gpu.log | cut -c1-300
for A in "--variant 30" "--variant 30 --tile 23" "--variant 30" "--variant 30 --tile 22"; do
  N=$(echo $A | tr -d ' -')
  timeout 300 python bench.py --mode fp32 $A --steps 10 --warmup 2 --layers --no-cpu-baseline --no-e2e --no-extras > $OUT/ab_fp32_$N.json 2> $OUT/ab_fp32_${N}_layers.txt
  echo "fp32 $A exit $? $(python -c "import json; d=json.loads(open('$OUT/ab_fp32_$N.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)" | tee -a $OUT/summary.txt
done
timeout 300 python bench.py --mode int8 --steps 10 --warmup 2 --layers --no-cpu-baseline --no-e2e --no-extras > $OUT/bench_int8.json 2> $OUT/bench_int8_layers.txt
echo "int8 exit $? $(python -c "import json; d=json.loads(open('$OUT/bench_int8.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)" | tee -a $OUT/summary.txt
timeout 900 python bench.py --layers > $OUT/bench.json 2> $OUT/bench_layers.txt
echo "bench exit $?" | tee -a $OUT/summary.txt
tail -1 $OUT/bench.json | cut -c1-300
du -sh $OUT
