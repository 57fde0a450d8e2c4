#!/bin/bash
# round 5: the evidence passes of the shipped tree -- rocprofv3 kernel stats (one configuration per run), FETCH_SIZE / WRITE_SIZE
# passes of the legs whose dominant kernel changed this round (fp32, tiny), matrix-pipe / wave-state / LDS counters of K1r and K1x
# in the network.  Usage: bash tools/r5_profile.sh <tag>
TAG=${1:-r5p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
export YL_HEAD_CACHE=/tmp/yl_head_cache
C1="--no-cpu-baseline --no-e2e --no-extras"
for leg in "c3_yolov3_608_b64_fp32|--mode fp32 --steps 7 --warmup 2" "c4_yolov3_608_b64_int8|--mode int8 --steps 10 --warmup 2" \
           "c2_yolov3_tiny_416_b32_fp32|--model yolov3-tiny --size 416 --batch 32 --mode fp32 --steps 20 --warmup 3"; do
  T=${leg%%|*}; A=${leg#*|}
  timeout 300 python $R/bench.py $A $C1 --steps 1 --warmup 0 > /dev/null 2>&1
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_$T -o s -- python $R/bench.py $A $C1 > $R/$OUT/stats_$T.json 2> $R/$OUT/stats_$T.err )
  echo "stats $T exit $?"
  F=$(find $OUT/stats_$T -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp "$F" $OUT/kernel_stats_$T.csv && head -8 "$F" | cut -c1-200
  tail -1 $OUT/stats_$T.json | cut -c1-200
done
find $OUT -name "*kernel_trace.csv" -delete
C2="--steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-extras --raw-head --nms 0"
for leg in fp32 tiny; do
  case $leg in
    fp32) A="--mode fp32";;
    tiny) A="--model yolov3-tiny --size 416 --batch 32 --mode fp32";;
  esac
  for C in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmcleg_${leg}/$C -o pmc -- python $R/bench.py $A $C2 > $R/$OUT/pmcleg_${leg}_$C.log 2>&1 )
    echo "pmcleg $leg $C exit $?"
  done
  python tools/pmc_summary.py $OUT/pmcleg_${leg} > $OUT/pmcleg_${leg}_summary.txt 2>&1
  cat $OUT/pmcleg_${leg}_summary.txt | head -40
done
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc_$N -o pmc -- python $R/bench.py --mode fp32 $C2 > $R/$OUT/pmc_$N.log 2>&1 )
  echo "pmc $N exit $?"
done
python tools/pmc_summary.py $OUT > $OUT/pmc_summary_all.txt 2>&1
grep -E "row3|x3" $OUT/pmc_summary_all.txt | head -60
find $OUT -name "*counter_collection.csv" -size +20M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
