#!/bin/bash
# round 5, GPU call: shader clock / power while ONE K1r layer runs back to back -- shipped build against lab builds without the A loads (128),
# without the input-row loads but with their transform + split (768 = 256 | 512), without any global load (896 = 128 | 256 | 512)
OUT=gpurun_out/${1:-r5l}; mkdir -p $OUT
for v in ${BUILDS:-0 128 768 896}; do
  echo "== build $v" | tee -a $OUT/clock_row3_builds.txt
  YOLO2HIP_LIB=$PWD/tools/ab/libyolo2hip_x$v.so PROBE_DELAY=9 PROBE_N=6 bash tools/clock_probe.sh $OUT/clk_$v.txt python tools/sweep_conv.py --batch 64 --tiles 61 --only ${SHAPE:-12} --iters 20000 --variant 0
  cat $OUT/clk_$v.txt | sed 's/GPU\[0\]\t\t: //g; s/=* Power Consumption =*;//' | tee -a $OUT/clock_row3_builds.txt
done
