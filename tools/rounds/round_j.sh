#!/bin/bash
OUT=gpurun_out/${1:-r1j}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 2 --layers > $OUT/bench.json 2> $OUT/bench_layers.txt; echo "fp32 exit $?"
timeout 600 python bench.py --mode int8 --steps 10 --warmup 2 --layers > $OUT/bench_int8.json 2> $OUT/bench_int8_layers.txt; echo "int8 exit $?"
timeout 600 python bench.py --mode int8 --steps 10 --warmup 2 --no-cpu-baseline --no-fuse > $OUT/bench_int8_nofuse.json 2>/dev/null; echo "int8 nofuse exit $?"
python - <<'PY'
import json,os
root=os.environ.get('GRAFT_REPO_ROOT','.')+"/gpurun_out/"+os.environ.get('TAG','r1j')
for t in ["bench","bench_int8","bench_int8_nofuse"]:
    d=json.load(open(root+"/%s.json"%t))
    r=d["roofline"]; print(t, "%.1f img/s %.2f ms | dom %s %.1f TF | conv %.2f other %.2f"%(d["value"],d["ms_per_step"],r["kernel"],r["achieved"],r["conv_ms_per_step"],r["other_layers_ms_per_step"]), {k:round(v["ms_per_step"],2) for k,v in r["by_kernel"].items()}, "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
