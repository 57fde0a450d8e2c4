#!/bin/bash
# K1r view form (tile 70 / variant bit 13) against the shipped tiles: parity tests, per-layer sweep, in-network A/B
TAG=${1:-r5v}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_row3.py -x -q > $OUT/pytest_row3.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_row3.log
tail -3 $OUT/pytest_row3.log
DEF=$(python -c "import sys; sys.path.insert(0, 'tests'); import common; print(common.VARIANT_DEFAULT)")
for v in $DEF $((DEF | 8192)) $DEF $((DEF | 8192)); do
  timeout 600 python bench.py --mode fp32 --no-extras --no-cpu-baseline --no-e2e --steps 10 --warmup 3 --variant $v --layers > $OUT/bench_v$v.json 2>$OUT/bench_v$v.err
  python - <<PY | tee -a $OUT/bench.txt
import json
try:
    r = json.loads(open("$OUT/bench_v$v.json").read().strip().splitlines()[-1])
    bk = r["roofline"]["by_kernel"]
    print("variant $v: %.1f img/s" % r["value"], {k: (round(x["ms_per_step"], 3), x["launches"]) for k, x in bk.items() if "row3" in k})
except Exception as e:
    print("variant $v: failed", e)
PY
done
