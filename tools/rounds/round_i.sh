#!/bin/bash
OUT=gpurun_out/${1:-r1i}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --layers > $OUT/bench.json 2> $OUT/bench_layers.txt; echo "fp32 exit $?"
timeout 600 python bench.py --mode int8 --steps 10 --warmup 2 --no-cpu-baseline --layers > $OUT/bench_int8_a.json 2> $OUT/bench_int8_a_layers.txt; echo "A exit $?"
YL_I8_TILE=64 timeout 600 python bench.py --mode int8 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_int8_b.json 2>/dev/null; echo "B exit $?"
timeout 600 python bench.py --mode int8 --steps 10 --warmup 2 --no-cpu-baseline --no-fuse > $OUT/bench_int8_c.json 2>/dev/null; echo "C exit $?"
timeout 900 python tools/sweep_conv.py --batch 64 --iters 5 --only 0,1,2,3,5,8,9,12,15,14 > $OUT/sweep.txt 2>&1; echo "sweep exit $?"; grep "^#" $OUT/sweep.txt
python - <<'PY'
import json,os
root=os.environ.get('GRAFT_REPO_ROOT','.')+"/gpurun_out/"+os.environ.get('TAG','r1i')
for t in ["bench","bench_int8_a","bench_int8_b","bench_int8_c"]:
    d=json.load(open(root+"/%s.json"%t))
    r=d["roofline"]; print(t, "%.1f img/s %.2f ms | dom %.1f TF | conv %.2f other %.2f"%(d["value"],d["ms_per_step"],r["achieved"],r["conv_ms_per_step"],r["other_layers_ms_per_step"]), {k:round(v["ms_per_step"],2) for k,v in r["by_kernel"].items()})
PY
