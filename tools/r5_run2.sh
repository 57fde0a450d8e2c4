#!/bin/bash
# round 5, GPU call 2: where K1r's wrong outputs sit (tools/r5_diag_row3.py) + what bounds the kernel (lab builds, tools/ab_builds.sh)
TAG=${1:-r5b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python tools/r5_diag_row3.py 61,62,68 > $OUT/diag_row3.txt 2>&1
echo "diag exit $?"
cat $OUT/diag_row3.txt | cut -c1-400
ABFILE=conv_f32_row3 TILES=61,62 SHAPES=9,12 ITERS=30 timeout 900 bash tools/ab_builds.sh run "0 1 2 4 6 8 16 32 64" 0 > $OUT/ablation_row3.txt 2>&1
echo "ablation exit $?"
cat $OUT/ablation_row3.txt
