#!/bin/bash
# round 3: K1f (VALU first-layer kernel) -- targeted parity + the benches whose first layer it replaces
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3first; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_int8_xnor.py -m gpu -x -q -k "${KSEL:-xnor}" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $O/pytest_sel.log
tail -5 $O/pytest_sel.log
timeout 300 python bench.py --model tiny-yolo-xnor --size 416 --batch 128 --mode fp32 --no-extras --no-e2e --no-cpu-baseline --layers > $O/bench_xnor.json 2> $O/bench_xnor.err; tail -1 $O/bench_xnor.json | cut -c1-200
timeout 300 python bench.py --model yolov3-tiny --size 416 --batch 32 --mode fp32 --no-extras --no-e2e --no-cpu-baseline --layers > $O/bench_tiny.json 2> $O/bench_tiny.err; tail -1 $O/bench_tiny.json | cut -c1-200
if [ "${FULL608:-0}" = 1 ]; then
timeout 300 python bench.py --no-extras --no-e2e --no-cpu-baseline --layers > $O/bench_608.json 2> $O/bench_608.err; tail -1 $O/bench_608.json | cut -c1-200
fi
grep "conv_xnor\|type= 3" $O/bench_xnor.err | head -20
grep -i "conv_f32_first\|smallk" $O/bench_*.err | head
