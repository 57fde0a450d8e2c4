#!/bin/bash
# round 5: K1x pinned schedule in the network (default) against variant bit 12 (hipcc's own order), alternating, same box
OUT=gpurun_out/${1:-r5n}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -k "x3" -m gpu -q --maxfail=5 > $OUT/pytest_x3.log 2>&1
echo "pytest exit $?"; tail -3 $OUT/pytest_x3.log
export YL_HEAD_CACHE=/tmp/yl_head_cache
C1="--mode fp32 --no-cpu-baseline --no-e2e --no-extras --steps 10 --warmup 3"
for leg in "pinned|" "plain|--variant 7230" "pinned2|" "plain2|--variant 7230"; do
  T=${leg%%|*}; A=${leg#*|}
  timeout 300 python bench.py $C1 $A > $OUT/bench_$T.json 2> $OUT/bench_$T.err
  echo "bench $T $(tail -1 $OUT/bench_$T.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["value"],1), "img/s", {k:(v["launches"],round(v["ms_per_step"],3)) for k,v in r["by_kernel"].items() if "x3" in k})' 2>&1 | tail -1)" | tee -a $OUT/ab_x3_pinned.txt
done
