#!/bin/bash
# round 5, GPU call 1: K1r (conv_f32_row3.hip) -- parity of every tile, accuracy against float64, per-shape sweep against the
# 2-D FP32 Winograd kernel, whole-network A/B.  Usage: bash tools/r5_run1.sh <tag>
TAG=${1:-r5a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $OUT/device.txt
echo "== row3 tests" | tee $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_row3.py -m gpu -q -x --durations=5 > $OUT/pytest_row3.log 2>&1
echo "row3 tests exit $?" | tee -a $OUT/summary.txt
tail -15 $OUT/pytest_row3.log
echo "== sweep" | tee -a $OUT/summary.txt
timeout 500 python tools/sweep_conv.py --batch 64 --tiles 31,61,62,63,64,65,66,68 --iters 30 --only 3,6,9,12,15 > $OUT/sweep_row3_b64.txt 2>&1
echo "sweep exit $?" | tee -a $OUT/summary.txt
grep "^#" $OUT/sweep_row3_b64.txt
python - "$OUT/sweep_row3_b64.txt" <<'PY' >> $OUT/summary.txt
import json,sys
rows=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
for r in rows: print("shape %2d M=%4d C=%4d H=%3d tile %2d %-34s %.3f ms" % (r["shape"],r["M"],r["C"],r["H"],r["tile"],r["kernel"],r["ms"]))
PY
C1="--mode fp32 --no-cpu-baseline --no-e2e --no-extras --steps 10 --warmup 3"
export YL_HEAD_CACHE=/tmp/yl_head_cache
for leg in "default|--layers" "r4default|--variant 1086 --layers" "tile62|--tile 62" "tile64|--tile 64" "tile63|--tile 63" "default2|"; do
  T=${leg%%|*}; A=${leg#*|}
  timeout 300 python bench.py $C1 $A > $OUT/bench_$T.json 2> $OUT/bench_$T.err
  echo "bench $T exit $? $(tail -1 $OUT/bench_$T.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", d["roofline"]["kernel"], round(d["roofline"]["avg_launch_ms"],4), "ms frac", round(d["roofline"]["frac"],3))' 2>&1 | tail -1)" | tee -a $OUT/summary.txt
done
echo "== float64 truth" | tee -a $OUT/summary.txt
timeout 900 python -m pytest "tests/test_gpu_parity.py::test_fp32_error_vs_float64_truth" -m gpu -q -s > $OUT/pytest_truth.log 2>&1
echo "truth exit $?" | tee -a $OUT/summary.txt
grep -E "worst|head [0-9]|passed|failed|Error|assert" $OUT/pytest_truth.log | head -30 | tee -a $OUT/summary.txt
cat $OUT/summary.txt
