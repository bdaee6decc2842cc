#!/bin/bash
# PMC passes over the LBS kernel (run on the GPU box): tools/pmc_lbs.sh TAG FRAMES
set -e
TAG=$1; FR=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcl_$TAG
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dev_lbs_time.py - $FR > $OUT.p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT.p$i.log; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'lbs_mfma' in r['Kernel_Name'] and int(r['Grid_Size']) > 100000:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc): print(f"{k:32s} {sum(acc[k])/len(acc[k]):16.0f}  (n={len(acc[k])})")
PY
