#!/bin/bash
# INT8: [upsample] folded into the quantise pass of the [route] behind it -- tests, then the INT8 leg (per-layer table)
OUT=gpurun_out/${1:-r5u}; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_int8_xnor.py tests/test_gpu_headline.py -x -q -k "int8 or quant or xnor or first" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for rep in 1 2; do
timeout 600 python bench.py --mode int8 --no-cpu-baseline --no-e2e --steps 20 --warmup 3 --layers --no-extras > $OUT/bench_int8_$rep.json 2>$OUT/bench_int8_$rep.err
python - <<PY | tee -a $OUT/bench.txt
import json
r = json.loads(open("$OUT/bench_int8_$rep.json").read().strip().splitlines()[-1])
print("int8: %.1f img/s %.3f ms/step detect %.3f" % (r["value"], r["ms_per_step"], r.get("detect_ms_per_step") or -1))
PY
grep -E "^ *(8[4-9]|9[0-9]) type" $OUT/bench_int8_$rep.err | head -16
done
