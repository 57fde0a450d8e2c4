#!/bin/bash
OUT=gpurun_out/${1:-r1h}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_int8_xnor.py tests/test_gpu_dropin.py tests/test_golden.py -m gpu -q -x > $OUT/pytest_int8.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_int8.log
timeout 600 python bench.py --mode int8 --steps 10 --warmup 2 --no-cpu-baseline --layers > $OUT/bench_int8_a.json 2> $OUT/bench_int8_a_layers.txt; echo "A exit $?"
YL_I8_TILE=64 timeout 600 python bench.py --mode int8 --steps 10 --warmup 2 --no-cpu-baseline --layers > $OUT/bench_int8_b.json 2> $OUT/bench_int8_b_layers.txt; echo "B exit $?"
timeout 600 python bench.py --mode int8 --steps 10 --warmup 2 --no-cpu-baseline --no-fuse > $OUT/bench_int8_c.json 2>/dev/null; echo "C exit $?"
python - <<'PY'
import json,os
for t in "abc":
    d=json.load(open(os.environ.get('GRAFT_REPO_ROOT','.')+"/gpurun_out/%s/bench_int8_%s.json"%(os.environ.get('TAG','r1h'),t)))
    r=d["roofline"]; print(t, "%.1f img/s %.2f ms | conv %.2f other %.2f"%(d["value"],d["ms_per_step"],r["conv_ms_per_step"],r["other_layers_ms_per_step"]), {k:round(v["ms_per_step"],2) for k,v in r["by_kernel"].items()})
PY
