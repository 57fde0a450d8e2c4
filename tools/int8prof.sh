#!/bin/bash
# PMC passes over the -quantized bench (one step, batch 64): where do the INT8 kernels spend time?
OUT=gpurun_out/${1:-r1g}; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/i8pmc_$N -o pmc -- python $R/bench.py --mode int8 --steps 1 --warmup 0 --no-cpu-baseline > $R/$OUT/i8pmc_$N.log 2>&1 )
  echo "pmc $N exit $?"
done
python - <<'PY'
import csv,glob,os,collections
root=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/'+(os.environ.get('TAG') or 'r1g')
acc=collections.defaultdict(lambda:[0,0.0]); dur=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob(root+'/i8pmc_*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']; k=k[:k.find('(')].replace('void ','').replace('yl::','')[:44]
        a=acc[(k,r['Counter_Name'])]; a[0]+=1; a[1]+=float(r['Counter_Value'])
        if r['Counter_Name'] in('SQ_WAVE_CYCLES','FETCH_SIZE'):
            d=dur[(k,r['Counter_Name'])]; d[0]+=1; d[1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
for (k,c),(n,s) in sorted(acc.items()):
    if 'i8' in k or 'quantize' in k: print("%-46s %-26s n=%3d mean %14.1f"%(k,c,n,s/n))
for (k,c),(n,s) in sorted(dur.items()):
    if 'i8' in k or 'quantize' in k: print("DUR %-46s (%s pass) n=%3d mean %.1f us"%(k,c,n,s/n))
PY
find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
