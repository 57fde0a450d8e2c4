#!/bin/bash
# K1m (first layer on the FP32 matrix pipe, variant bit 14) against K1f: the tests that pin its bits, then the legs it serves (XNOR config 5, INT8)
TAG=${1:-r5f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
DEF=$(python -c "import sys; sys.path.insert(0, 'tests'); import common; print(common.VARIANT_DEFAULT)")
OFF=$((DEF & ~16384))
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_int8_xnor.py -x -q -k "first or sign or fusion or xnor" > $OUT/pytest_first.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_first.log
tail -3 $OUT/pytest_first.log
for v in $OFF $DEF $OFF $DEF; do
timeout 600 python bench.py --model tiny-yolo-xnor --size 416 --batch 128 --mode fp32 --no-extras --no-cpu-baseline --no-e2e --steps 30 --warmup 5 --layers --variant $v > $OUT/bench_xnor.json 2>$OUT/bench_xnor.err
timeout 600 python bench.py --mode int8 --no-extras --no-cpu-baseline --no-e2e --steps 20 --warmup 3 --layers --variant $v > $OUT/bench_int8.json 2>$OUT/bench_int8.err
for leg in xnor int8; do
  python - <<PY | tee -a $OUT/bench.txt
import json
try:
    r = json.loads(open("$OUT/bench_$leg.json").read().strip().splitlines()[-1])
    first = [l.strip() for l in open("$OUT/bench_$leg.err") if "conv_f32_first" in l]
    print("variant $v $leg: %.1f img/s %.3f ms/step |" % (r["value"], r["ms_per_step"]), first[:1])
except Exception as e:
    print("variant $v $leg: failed", e)
PY
done
done
