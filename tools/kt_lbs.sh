#!/bin/bash
# rocprofv3 kernel times of the LBS launches for several library variants (run on the GPU box):
#   tools/kt_lbs.sh FRAMES[:smplx] name...      ("-" = the product library); prints the average duration per kernel
FR=${1%%:*}; KIND=smpl; [[ "$1" == *:* ]] && KIND=${1#*:}; shift
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  D=/tmp/kt_$n_$$_$RANDOM
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/dev_lbs_time.py $n $FR $KIND > $D.log 2>&1 || { echo "$n failed"; tail -3 $D.log; continue; }
  python3 - "$D" "$n" "$FR" "$KIND" <<'PY'
import csv, glob, sys
d, n, fr, kind = sys.argv[1:5]
for f in glob.glob(d + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "lbs" in r["Name"] or "pose_setup" in r["Name"]:
            print(f"{n:10s} {kind} {fr}: {r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.2f} us  min {float(r['MinNs'])/1e3:9.2f}")
PY
done
