#!/bin/bash
# Round-2 GPU call H: final validation -- parity suite (uncaptured, so a runtime abort would leave its message),
# smoke, default bench line, rocprofv3 kernel stats of the FP32 and INT8 legs
TAG=${1:-r2h}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider --maxfail=12 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/summary.txt; grep -n -i "fault\|abort\|terminate\|free()\|corrupt" $OUT/pytest_gpu.log | head -5; tail -6 $OUT/pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/summary.txt; tail -1 $OUT/smoke.log
timeout 900 python bench.py --layers > $OUT/bench.json 2> $OUT/bench_layers.txt
echo "bench exit $?" | tee -a $OUT/summary.txt
tail -1 $OUT/bench.json | cut -c1-300
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/rocprof -o bench -- python $R/bench.py --mode fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extras --raw-head --nms 0 > $R/$OUT/rocprof_run.log 2>&1 )
echo "rocprof exit $?" | tee -a $OUT/summary.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/rocprof_i8 -o bench -- python $R/bench.py --mode int8 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extras --raw-head --nms 0 > $R/$OUT/rocprof_i8_run.log 2>&1 )
echo "rocprof int8 exit $?" | tee -a $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -size +5M -delete
du -sh $OUT
