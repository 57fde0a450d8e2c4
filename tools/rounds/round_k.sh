#!/bin/bash
OUT=gpurun_out/${1:-r1k}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_gpu.log
timeout 600 python bench.py --model tiny-yolo-xnor --size 416 --batch 128 --steps 20 --warmup 3 --layers --no-cpu-baseline > $OUT/bench_xnor.json 2> $OUT/bench_xnor_layers.txt; echo "xnor exit $?"
timeout 600 python bench.py --model yolov3-tiny --size 416 --batch 32 --steps 20 --warmup 3 --layers > $OUT/bench_tiny.json 2> $OUT/bench_tiny_layers.txt; echo "tiny exit $?"
grep -v amdgpu $OUT/bench_xnor_layers.txt; grep -v amdgpu $OUT/bench_tiny_layers.txt
python - <<'PY'
import json,os
root=os.environ.get('GRAFT_REPO_ROOT','.')+"/gpurun_out/"+os.environ.get('TAG','r1k')
for t in ["bench_xnor","bench_tiny"]:
    d=json.load(open(root+"/%s.json"%t))
    r=d["roofline"]; print(t, "%.1f img/s %.3f ms | conv %.2f other %.2f"%(d["value"],d["ms_per_step"],r["conv_ms_per_step"],r["other_layers_ms_per_step"]), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
