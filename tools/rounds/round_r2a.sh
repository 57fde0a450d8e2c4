#!/bin/bash
# Round-2 GPU call A: the new FP32 schedule variants A/B in the network (same box), then the full parity suite,
# the default bench line and the rocprofv3 kernel stats.  Usage: bash tools/round_r2a.sh <tag>
TAG=${1:-r2a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== new tests first (fail fast)" | tee -a $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "variants or float4 or smallk" > $OUT/pytest_new.log 2>&1
echo "pytest new exit $?" | tee -a $OUT/summary.txt; tail -15 $OUT/pytest_new.log
timeout 300 python -m pytest tests/test_gpu_int8_xnor.py -m gpu -q -x -k "first_layer_kernel" >> $OUT/pytest_new.log 2>&1
echo "pytest new int8 exit $?" | tee -a $OUT/summary.txt; tail -5 $OUT/pytest_new.log
echo "== FP32 variants A/B" | tee -a $OUT/summary.txt
for V in 0 1 2 3 4 8 15 0; do
  timeout 300 python bench.py --mode fp32 --variant $V --steps 10 --warmup 2 --layers --no-cpu-baseline --no-e2e --no-extras > $OUT/ab_fp32_v$V.json 2> $OUT/ab_fp32_v${V}_layers.txt
  echo "fp32 variant $V exit $? $(python -c "import json; d=json.loads(open('$OUT/ab_fp32_v$V.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)" | tee -a $OUT/summary.txt
done
for V in 0 8; do
  timeout 300 python bench.py --mode int8 --variant $V --steps 10 --warmup 2 --layers --no-cpu-baseline --no-e2e --no-extras > $OUT/ab_int8_v$V.json 2> $OUT/ab_int8_v${V}_layers.txt
  echo "int8 variant $V exit $? $(python -c "import json; d=json.loads(open('$OUT/ab_int8_v$V.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)" | tee -a $OUT/summary.txt
done
echo "== pytest -m gpu (all)" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --durations=8 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -25 $OUT/pytest_gpu.log
echo "== default bench line" | tee -a $OUT/summary.txt
timeout 900 python bench.py --layers > $OUT/bench.json 2> $OUT/bench_layers.txt
echo "bench exit $?" | tee -a $OUT/summary.txt
tail -1 $OUT/bench.json | cut -c1-600
echo "== rocprofv3 kernel stats" | tee -a $OUT/summary.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/rocprof -o bench -- python $R/bench.py --mode fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extras --raw-head --nms 0 > $R/$OUT/rocprof_run.log 2>&1 )
echo "rocprof exit $?" | tee -a $OUT/summary.txt
F=$(find $OUT/rocprof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -14 "$F" | cut -c1-200
find $OUT/rocprof -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
