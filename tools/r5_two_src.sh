#!/bin/bash
# FP32: [upsample] -> [route] -> conv 1x1 read from the two sources by K1x -- tests, then the FP32 leg A/B against fusion of the
# previous form (variant bit 12 = K1x without pinned schedule also switches the two-source form off: not a clean A/B; use the per-layer table)
OUT=gpurun_out/${1:-r5t}; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_row3.py tests/test_gpu_parity.py tests/test_gpu_headline.py -x -q -k "whole_network or fusion or fused or batch1" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for rep in 1 2; do
timeout 600 python bench.py --mode fp32 --no-cpu-baseline --no-e2e --steps 10 --warmup 3 --layers --no-extras > $OUT/bench_fp32_$rep.json 2>$OUT/bench_fp32_$rep.err
python - <<PY | tee -a $OUT/bench.txt
import json
r = json.loads(open("$OUT/bench_fp32_$rep.json").read().strip().splitlines()[-1])
print("fp32: %.1f img/s %.3f ms/step" % (r["value"], r["ms_per_step"]))
PY
grep -E "^ *(8[4-7]|9[6-9]) type" $OUT/bench_fp32_$rep.err | head -8
done
