#!/bin/bash
# PMC passes over the LBS vertex kernel (stream / tile) (run on the GPU box): tools/pmc_tile.sh TAG FRAMES [LIB] [smpl|smplx]
# Counters in separate passes (kernel-trace only beside --pmc), summary printed and written to gpurun_out/pmct_TAG.txt
TAG=$1; FR=$2; LIB=${3:--}; KIND=${4:-smpl}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmct_$TAG
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dev_lbs_time.py $LIB $FR $KIND > $OUT.p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT.p$i.log; }
done
python3 - <<PY | tee $OUT.txt
import csv, glob, collections
acc = collections.defaultdict(list)
dur = []
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if ('lbs_tile' in r['Kernel_Name'] or 'lbs_stream' in r['Kernel_Name']) and int(r['Grid_Size']) > 100000:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob("$OUT/p*/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if ('lbs_tile' in r['Kernel_Name'] or 'lbs_stream' in r['Kernel_Name']) and int(r['Grid_Size_X']) > 100000:
            dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print("frames $FR $KIND lib $LIB: vertex kernel launches", len(dur), "avg", sum(dur) / max(1, len(dur)) / 1e3, "us")
for k in sorted(acc): print(f"{k:32s} {sum(acc[k])/len(acc[k]):16.0f}  (n={len(acc[k])})")
PY
