#!/bin/bash
# round 5: counters of K1m (conv_f32_firstm.hip) inside config 5 (tiny-yolo-xnor 416 b128) and config 4 (yolov3 608 b64 INT8):
# HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes), matrix-pipe busy, instruction mix
TAG=${1:-r5q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
export YL_HEAD_CACHE=/tmp/yl_head_cache
C2="--steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-extras --raw-head --nms 0"
for leg in xnor int8; do
  case $leg in
    xnor) A="--model tiny-yolo-xnor --size 416 --batch 128 --mode fp32";;
    int8) A="--mode int8";;
  esac
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-32)
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc_${leg}/$N -o pmc -- python $R/bench.py $A $C2 > $R/$OUT/pmc_${leg}_$N.log 2>&1 )
    echo "pmc $leg $N exit $?"
  done
  python tools/pmc_summary.py $OUT/pmc_${leg} | grep -E "^#|kernel|first" > $OUT/pmc_${leg}_first_layer.txt
  cat $OUT/pmc_${leg}_first_layer.txt
  rm -rf $OUT/pmc_${leg}
done
