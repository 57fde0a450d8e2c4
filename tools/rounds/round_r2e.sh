#!/bin/bash
# Round-2 GPU call E: parity suite, PMC counter passes for the three dominant kernels (FP32 Winograd, INT8, XNOR),
# default bench line + rocprofv3 kernel stats.  Counter passes are their own runs with --kernel-trace only.
TAG=${1:-r2e}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --durations=5 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/summary.txt; tail -12 $OUT/pytest_gpu.log | cut -c1-300
pmc_pass() {   # tag, bench args...
  local T=$1; shift
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-24)
    ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc_${T}/$N -o pmc -- python $R/bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-extras --raw-head --nms 0 > $R/$OUT/pmc_${T}_$N.log 2>&1 )
    echo "pmc $T $N exit $?" | tee -a $OUT/summary.txt
  done
  python tools/pmc_summary.py $OUT/pmc_$T > $OUT/pmc_${T}_summary.txt 2>&1
  find $OUT/pmc_$T -name "*.csv" -size +1M -delete
}
pmc_pass fp32 --mode fp32
pmc_pass int8 --mode int8
pmc_pass xnor --model tiny-yolo-xnor --size 416 --batch 128
timeout 900 python bench.py --layers > $OUT/bench.json 2> $OUT/bench_layers.txt
echo "bench exit $?" | tee -a $OUT/summary.txt
tail -1 $OUT/bench.json | cut -c1-300
timeout 300 python bench.py --model tiny-yolo-xnor --size 416 --batch 128 --steps 20 --warmup 3 --layers --no-cpu-baseline > $OUT/bench_xnor.json 2> $OUT/bench_xnor_layers.txt
echo "xnor exit $?" | tee -a $OUT/summary.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/rocprof -o bench -- python $R/bench.py --mode fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extras --raw-head --nms 0 > $R/$OUT/rocprof_run.log 2>&1 )
echo "rocprof exit $?" | tee -a $OUT/summary.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/rocprof_i8 -o bench -- python $R/bench.py --mode int8 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extras --raw-head --nms 0 > $R/$OUT/rocprof_i8_run.log 2>&1 )
echo "rocprof int8 exit $?" | tee -a $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -size +5M -delete
du -sh $OUT
