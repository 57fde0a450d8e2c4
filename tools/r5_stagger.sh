#!/bin/bash
# phase stagger of K1r launches (variant bit 14): dispatcher map, parity of the row3 tests with it on, in-network A/B
TAG=${1:-r5s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
build/hwid_probe 2000 > $OUT/hwid_probe.txt 2>&1; head -3 $OUT/hwid_probe.txt
for v in ${VARIANTS:-3134 19518 3134 19518 27710}; do
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
