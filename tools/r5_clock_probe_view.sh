#!/bin/bash
# round 5, GPU call: shader clock / power (a) while the whole FP32 step runs, K1r pinned tile against its view form (variant bit 13),
# (b) while ONE K1r layer runs back to back, tile 61 against tile 70
OUT=gpurun_out/${1:-r5w}; mkdir -p $OUT
DEF=$(python -c "import sys; sys.path.insert(0, 'tests'); import common; print(common.VARIANT_DEFAULT)")
for v in $DEF $((DEF | 8192)); do
  echo "== step, variant $v" | tee -a $OUT/clock_view.txt
  PROBE_DELAY=12 PROBE_N=8 bash tools/clock_probe.sh $OUT/clk_step_$v.txt python bench.py --mode fp32 --no-extras --no-cpu-baseline --no-e2e --steps 400 --warmup 3 --variant $v
  cat $OUT/clk_step_$v.txt | sed 's/GPU\[0\]\t\t: //g; s/=* Power Consumption =*;//' | cut -c1-260 | tee -a $OUT/clock_view.txt
done
for t in 61 70; do
  for sh in 9 12; do
  echo "== layer $sh, tile $t" | tee -a $OUT/clock_view.txt
  PROBE_DELAY=9 PROBE_N=6 bash tools/clock_probe.sh $OUT/clk_${sh}_$t.txt python tools/sweep_conv.py --batch 64 --tiles $t --only $sh --iters 15000 --variant 0
  cat $OUT/clk_${sh}_$t.txt | sed 's/GPU\[0\]\t\t: //g; s/=* Power Consumption =*;//' | tee -a $OUT/clock_view.txt
  done
done
