#!/bin/bash
# Round-2 GPU call I: validation of the reverted (r2f-state) kernels: full async parity suite + default bench line
TAG=${1:-r2i}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -s -p no:cacheprovider --maxfail=12 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/summary.txt; grep -n -i "fault\|abort" $OUT/pytest_gpu.log | head -5; tail -4 $OUT/pytest_gpu.log | cut -c1-300
timeout 400 python bench.py --layers > $OUT/bench.json 2> $OUT/bench_layers.txt
echo "bench exit $?" | tee -a $OUT/summary.txt
tail -1 $OUT/bench.json | cut -c1-300
