#!/bin/bash
# PMC passes over the fit kernel (run on the GPU box): tools/pmc_fit.sh TAG FRAMES [debug_launch_shape]
set -e
TAG=$1; FR=$2; SHAPE=${3:-0}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dev_fit_once.py $FR - 3 - $SHAPE > $OUT.p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT.p$i.log; }
done
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(list)
raw = []
for f in sorted(glob.glob("$OUT/p*/*/*counter_collection.csv")):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'fit_world' in r['Kernel_Name'] or 'fit_tree' in r['Kernel_Name']:
            per[r['Counter_Name']].append(r)
    for name, rows in per.items():
        for r in rows[-8:]:                          # the last eight dispatches of the pass (steady clock), every counter
            acc[name].append(float(r['Counter_Value']))
            raw.append((r['Kernel_Name'].split('(')[0], r['Grid_Size'], r['Workgroup_Size'], r['VGPR_Count'], name, r['Counter_Value']))
with open("$OUT.raw.csv", "w") as f:                 # every dispatch of the fit kernel in every pass: what profiles/ keeps
    f.write("kernel,grid_size,workgroup_size,vgpr_count,counter,value\n")
    for row in raw: f.write(",".join(row) + "\n")
avg = {k: sum(v) / len(v) for k, v in acc.items()}
json.dump({"frames": $FR, "per_launch_average": avg, "dispatches": {k: len(v) for k, v in acc.items()}}, open("$OUT.summary.json", "w"), indent=1)
for k in sorted(avg): print(f"{k:32s} {avg[k]:16.0f}  (n={len(acc[k])})")
PY
