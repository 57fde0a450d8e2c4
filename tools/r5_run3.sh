#!/bin/bash
# round 5, GPU call 3: the whole GPU suite on the K1r default + the default bench line (no CPU baseline)
TAG=${1:-r5c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --durations=8 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"
tail -45 $OUT/pytest_gpu.log | cut -c1-250
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --layers > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench exit $?"
tail -1 $OUT/bench_default.json | python -c '
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print("value", round(d["value"],1), "int8", round(d["int8"]["value"],1), "dominant", r["kernel"], round(r["avg_launch_ms"],4), "frac", round(r["frac"],3))
print("per_pipe", json.dumps(r["per_pipe"]))
print("batch_sweep", {k: round(v["images_per_sec"],1) if isinstance(v,dict) and "images_per_sec" in v else v for k,v in d.get("batch_sweep",{}).items()})
for k in ("config2_yolov3_tiny_416_b32_fp32","config5_tiny_yolo_xnor_416_b128","group_n1","torchrun_world1"):
    v=d.get(k,{}); print(k, v.get("value"), v.get("error"))
print("by_kernel", json.dumps({k:(round(v["ms_per_step"],3), v["launches"]) for k,v in r["by_kernel"].items()}))
'
