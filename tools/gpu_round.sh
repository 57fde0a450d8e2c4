#!/bin/bash
# One gpurun invocation: parity tests, smoke, bench (+ per-layer table), rocprofv3 kernel stats.
# Usage (from the repo root on the GPU box):  bash tools/gpu_round.sh [tag]
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $OUT/device.txt
lscpu | grep -E "Model name|^CPU\(s\)" >> $OUT/device.txt
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -x --durations=8 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -25 $OUT/pytest_gpu.log
echo "== smoke" | tee -a $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" | tee -a $OUT/summary.txt
tail -3 $OUT/smoke.log
echo "== bench yolov3 608 b64" | tee -a $OUT/summary.txt
timeout 900 python bench.py --steps 5 --warmup 2 --layers > $OUT/bench.json 2> $OUT/bench_layers.txt
echo "bench exit $?" | tee -a $OUT/summary.txt
cat $OUT/bench.json
tail -5 $OUT/bench_layers.txt
echo "== rocprofv3 kernel stats" | tee -a $OUT/summary.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/rocprof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof_run.log 2>&1 )
echo "rocprof exit $?" | tee -a $OUT/summary.txt
find $OUT/rocprof -name "*stats*" | head
F=$(find $OUT/rocprof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -15 "$F"
# keep the merged-back payload small: drop the raw per-dispatch trace
find $OUT/rocprof -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
