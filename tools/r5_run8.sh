#!/bin/bash
# round 5, GPU call 8: is the ping-pong structure fast when its staging requests cost nothing?  lab builds of commit 7604f15
OUT=gpurun_out/${1:-r5i}; mkdir -p $OUT
ABFILE=conv_f32_row3 TILES=61,63 SHAPES=9,12,15 ITERS=30 timeout 600 bash tools/ab_builds.sh run "0 1" 0 > $OUT/ablation_pingpong.txt 2>&1
cat $OUT/ablation_pingpong.txt
