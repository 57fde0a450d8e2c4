#!/bin/bash
# One gpurun invocation: parity tests, smoke, bench (+ per-layer table), rocprofv3 kernel stats
# and PMC counter passes.  Usage (repo root on the GPU box):  bash tools/gpu_round.sh <tag> [steps...]
# steps: tests newtests smoke bench sweep prof pmc int8 xnor valu stats4 pmclegs ...   (default: tests smoke bench prof)
TAG=${1:-r1}
shift
STEPS=${@:-tests smoke bench prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $OUT/device.txt
lscpu | grep -E "Model name|^CPU\(s\)" >> $OUT/device.txt
has() { [[ " $STEPS " == *" $1 "* ]]; }

if has tests; then
  echo "== pytest -m gpu" | tee -a $OUT/summary.txt
  timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 --durations=8 > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit $?" | tee -a $OUT/summary.txt
  tail -30 $OUT/pytest_gpu.log
fi
if has newtests; then
  echo "== pytest new files: $NEWTESTS" | tee -a $OUT/summary.txt
  timeout 900 python -m pytest ${NEWTESTS:-tests/test_gpu_detect.py tests/test_gpu_input.py} -m gpu -q -s ${NEWK:+-k "$NEWK"} --maxfail=20 --durations=8 > $OUT/pytest_new.log 2>&1
  echo "pytest new exit $?" | tee -a $OUT/summary.txt
  tail -40 $OUT/pytest_new.log
fi
if has valu; then
  # VALU issue rate (wave-instructions per SIMD clock) of the instruction mixes the XNOR / first-layer roofs are quoted on
  echo "== VALU issue-rate microbenchmark" | tee -a $OUT/summary.txt
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue_bench tools/valu_issue_bench.hip 2> $OUT/valu_build.log
  timeout 120 /tmp/valu_issue_bench > $OUT/valu_issue_bench.txt 2>&1
  echo "valu exit $?" | tee -a $OUT/summary.txt
  cat $OUT/valu_issue_bench.txt
fi
if has stats4; then
  # single-configuration rocprofv3 kernel stats of the shipped tree: BASELINE configs 3 (FP32), 4 (INT8), 2, 5
  echo "== rocprofv3 --kernel-trace --stats, one configuration per run (--no-extras)" | tee -a $OUT/summary.txt
  # YL_HEAD_CACHE: the head-bias calibration of the synthetic weights (a batch-2 probe network) runs in an unprofiled pass
  # first, so that every launch in the CSV is a launch of the named configuration
  C1="--no-cpu-baseline --no-e2e --no-extras"
  export YL_HEAD_CACHE=/tmp/yl_head_cache
  for leg in "c3_yolov3_608_b64_fp32|--mode fp32 --steps 5 --warmup 2" "c4_yolov3_608_b64_int8|--mode int8 --steps 10 --warmup 2" \
             "c2_yolov3_tiny_416_b32_fp32|--model yolov3-tiny --size 416 --batch 32 --mode fp32 --steps 20 --warmup 3" \
             "c5_tiny_yolo_xnor_416_b128|--model tiny-yolo-xnor --size 416 --batch 128 --mode fp32 --steps 20 --warmup 3" \
             "bf16_yolov3_608_b64|--mode bf16 --steps 10 --warmup 2"; do
    T=${leg%%|*}; A=${leg#*|}
    timeout 300 python $R/bench.py $A $C1 --steps 1 --warmup 0 > /dev/null 2>&1
    ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_$T -o s -- python $R/bench.py $A $C1 > $R/$OUT/stats_$T.json 2> $R/$OUT/stats_$T.err )
    echo "stats $T exit $?" | tee -a $OUT/summary.txt
    F=$(find $OUT/stats_$T -name "*kernel_stats.csv" | head -1)
    [ -n "$F" ] && cp "$F" $OUT/kernel_stats_$T.csv && head -8 "$F" | cut -c1-180
    tail -1 $OUT/stats_$T.json | cut -c1-300
  done
  find $OUT -name "*kernel_trace.csv" -delete
fi
if has pmclegs; then
  # FETCH_SIZE / WRITE_SIZE in their own runs (kernel-trace only), one configuration per run, batch of the bench line
  echo "== rocprofv3 PMC FETCH_SIZE / WRITE_SIZE passes per leg (--no-extras)" | tee -a $OUT/summary.txt
  C1="--steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-extras --raw-head --nms 0"
  for leg in ${PMCLEGS:-fp32 bf16}; do
    case $leg in
      fp32) A="--mode fp32";; int8) A="--mode int8";; bf16) A="--mode bf16";;
      tiny) A="--model yolov3-tiny --size 416 --batch 32 --mode fp32";;
      xnor) A="--model tiny-yolo-xnor --size 416 --batch 128 --mode fp32";;
    esac
    for C in FETCH_SIZE WRITE_SIZE; do
      ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmcleg_${leg}/$C -o pmc -- python $R/bench.py $A $C1 > $R/$OUT/pmcleg_${leg}_$C.log 2>&1 )
      echo "pmcleg $leg $C exit $?" | tee -a $OUT/summary.txt
    done
    python tools/pmc_summary.py $OUT/pmcleg_${leg} > $OUT/pmcleg_${leg}_summary.txt 2>&1
    cat $OUT/pmcleg_${leg}_summary.txt | head -40
  done
  find $OUT -name "*counter_collection.csv" -size +30M -delete
  find $OUT -name "*kernel_trace.csv" -delete
fi
if has smoke; then
  echo "== smoke" | tee -a $OUT/summary.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
  echo "smoke exit $?" | tee -a $OUT/summary.txt
  tail -3 $OUT/smoke.log
fi
if has bench; then
  echo "== bench yolov3 608 b64 fp32" | tee -a $OUT/summary.txt
  timeout 1200 python bench.py --steps 10 --warmup 2 --layers > $OUT/bench.json 2> $OUT/bench_layers.txt
  echo "bench exit $?" | tee -a $OUT/summary.txt
  cat $OUT/bench.json
fi
if has int8; then
  echo "== bench yolov3 608 b64 int8" | tee -a $OUT/summary.txt
  timeout 900 python bench.py --mode int8 --steps 10 --warmup 2 --layers > $OUT/bench_int8.json 2> $OUT/bench_int8_layers.txt
  echo "bench int8 exit $?" | tee -a $OUT/summary.txt
  cat $OUT/bench_int8.json
fi
if has xnor; then
  echo "== bench tiny-yolo-xnor 416 b128" | tee -a $OUT/summary.txt
  timeout 900 python bench.py --model tiny-yolo-xnor --size 416 --batch 128 --steps 10 --warmup 2 --layers --no-cpu-baseline > $OUT/bench_xnor.json 2> $OUT/bench_xnor_layers.txt
  echo "bench xnor exit $?" | tee -a $OUT/summary.txt
  cat $OUT/bench_xnor.json
  echo "== bench yolov3-tiny 416 b32 fp32" | tee -a $OUT/summary.txt
  timeout 900 python bench.py --model yolov3-tiny --size 416 --batch 32 --steps 20 --warmup 3 --layers --no-cpu-baseline > $OUT/bench_tiny.json 2> $OUT/bench_tiny_layers.txt
  cat $OUT/bench_tiny.json
fi
if has sweep; then
  echo "== conv tile sweep" | tee -a $OUT/summary.txt
  timeout 1200 python tools/sweep_conv.py --batch 64 --iters 3 > $OUT/sweep.txt 2>&1
  echo "sweep exit $?" | tee -a $OUT/summary.txt
  grep "^#" $OUT/sweep.txt
fi
if has prof; then
  echo "== rocprofv3 kernel stats" | tee -a $OUT/summary.txt
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/rocprof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --raw-head --nms 0 > $R/$OUT/rocprof_run.log 2>&1 )
  echo "rocprof exit $?" | tee -a $OUT/summary.txt
  F=$(find $OUT/rocprof -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && head -14 "$F" | cut -c1-200
  find $OUT/rocprof -name "*kernel_trace.csv" -size +20M -delete
fi
if has pmc; then
  echo "== rocprofv3 PMC passes (own runs, kernel-trace only)" | tee -a $OUT/summary.txt
  rocprofv3 -L > $OUT/pmc_list.txt 2>&1
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc_$N -o pmc -- python $R/bench.py --steps 1 --warmup 0 --batch 16 --no-cpu-baseline --no-e2e --raw-head --nms 0 > $R/$OUT/pmc_$N.log 2>&1 )
    echo "pmc $N exit $?" | tee -a $OUT/summary.txt
  done
  python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
  cat $OUT/pmc_summary.txt | head -40
  find $OUT -name "*counter_collection.csv" -size +30M -delete
  find $OUT -name "*kernel_trace.csv" -size +20M -delete
fi
if has i8tiles; then
  echo "== INT8 tile sweep (per-layer tables)" | tee -a $OUT/summary.txt
  for T in ${I8TILES:-0 1 3 4 5}; do
    timeout 600 python bench.py --mode int8 --i8-tile $T --steps 8 --warmup 2 --layers --no-cpu-baseline --no-e2e > $OUT/bench_int8_t$T.json 2> $OUT/bench_int8_t${T}_layers.txt
    echo "i8 tile $T exit $? $(python -c "import json,sys; d=json.loads(open('$OUT/bench_int8_t$T.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)" | tee -a $OUT/summary.txt
  done
fi
if has blockpmc; then
  # per-dispatch counters of the residual-block micro network: BLOCK_ARGS="--mode int8 --C 512 --H 38" BLOCK_TAG=i8s4
  echo "== block micro network + PMC: $BLOCK_ARGS" | tee -a $OUT/summary.txt
  timeout 300 python tools/block_bench.py $BLOCK_ARGS > $OUT/block_${BLOCK_TAG}.txt 2>&1
  cat $OUT/block_${BLOCK_TAG}.txt | tail -14
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-30)
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/blk_${BLOCK_TAG}_$N -o pmc -- python $R/tools/block_bench.py $BLOCK_ARGS --iters 1 > $R/$OUT/blk_${BLOCK_TAG}_$N.log 2>&1 )
    echo "blockpmc $BLOCK_TAG $N exit $?" | tee -a $OUT/summary.txt
  done
  python tools/pmc_dispatch.py $OUT conv > $OUT/block_${BLOCK_TAG}_pmc.txt 2>&1
  find $OUT -name "*kernel_trace.csv" -size +20M -delete
fi
if has pmcx; then
  # PMC passes for another bench mode: PMCX_ARGS="--mode int8" PMCX_TAG=int8 (own runs, kernel-trace only)
  echo "== rocprofv3 PMC passes for: $PMCX_ARGS" | tee -a $OUT/summary.txt
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmcx_${PMCX_TAG}_$N -o pmc -- python $R/bench.py $PMCX_ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --raw-head --nms 0 > $R/$OUT/pmcx_${PMCX_TAG}_$N.log 2>&1 )
    echo "pmcx $PMCX_TAG $N exit $?" | tee -a $OUT/summary.txt
  done
  python tools/pmc_summary.py $OUT > $OUT/pmcx_${PMCX_TAG}_summary.txt 2>&1
  head -60 $OUT/pmcx_${PMCX_TAG}_summary.txt
  find $OUT -name "*counter_collection.csv" -size +30M -delete
  find $OUT -name "*kernel_trace.csv" -size +20M -delete
fi
if has dist1; then
  echo "== bench through torch.distributed.run, world 1 (RCCL all-gather path)" | tee -a $OUT/summary.txt
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --raw-head --nms 0 > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err
  echo "dist1 exit $?" | tee -a $OUT/summary.txt
  tail -2 $OUT/bench_dist1.json | cut -c1-400
  tail -3 $OUT/bench_dist1.err
fi
if has pmc64; then
  echo "== rocprofv3 HBM traffic counters at the bench batch (64)" | tee -a $OUT/summary.txt
  for C in "FETCH_SIZE" "WRITE_SIZE"; do
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc64_$C -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --raw-head --nms 0 > $R/$OUT/pmc64_$C.log 2>&1 )
    echo "pmc64 $C exit $?" | tee -a $OUT/summary.txt
  done
  python tools/pmc_summary.py $OUT > $OUT/pmc64_summary.txt 2>&1
  grep -E "FETCH_SIZE|WRITE_SIZE" $OUT/pmc64_summary.txt | head -20
  find $OUT -name "*kernel_trace.csv" -size +20M -delete
fi
du -sh $OUT
