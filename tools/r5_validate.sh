#!/bin/bash
# round 5: the driver's end-of-round sequence on one box -- pytest -m gpu, smoke, the default bench line (+ per-layer table)
TAG=${1:-r5v}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $OUT/device.txt
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --durations=8 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"
tail -15 $OUT/pytest_gpu.log | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?"; tail -2 $OUT/smoke.log
timeout 900 python bench.py --layers > $OUT/bench_default.json 2> $OUT/bench_layers.txt
echo "bench exit $?"
grep -E "^\{" $OUT/bench_default.json | tail -1 | python -c '
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print("value", round(d["value"],1), "ms", round(d["ms_per_step"],2), "int8", round(d["int8"]["value"],1), "dominant", r["kernel"], round(r["avg_launch_ms"],4), "frac", round(r["frac"],3), "sclk", r.get("sclk_mhz"), "frac_at_sclk", r.get("frac_at_sclk"))
print("traffic", r.get("traffic"), r.get("traffic_source","")[:80])
print("batch_sweep", {k: (round(v["images_per_sec"],1) if isinstance(v,dict) and "images_per_sec" in v else v) for k,v in d.get("batch_sweep",{}).items()})
for k in ("config2_yolov3_tiny_416_b32_fp32","config5_tiny_yolo_xnor_416_b128","group_n1","torchrun_world1","bf16"):
    v=d.get(k,{}); print(k, v.get("value"), v.get("error"))
print("cpu", d.get("cpu_baseline",{}).get("value"), (d.get("cpu_baseline",{}).get("int8") or {}).get("value"))
print("ref int8 vs ref fp32", ((d.get("cpu_baseline",{}).get("int8") or {}).get("hip_int8_vs_reference_int8") or {}).get("reference_int8_vs_reference_fp32"))
print("launch", d.get("launch"))
'
