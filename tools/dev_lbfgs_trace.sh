#!/bin/bash
# rocprofv3 kernel trace of single-frame calls (tools/dev_single_frame.py): kernel durations and the gaps between consecutive
# kernels of the L-BFGS rounds (run on the GPU box): tools/dev_lbfgs_trace.sh
cd /tmp && export TMPDIR=/tmp
D=/tmp/lbfgs_trace_$$
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/dev_single_frame.py > $D.log 2>&1 || { tail -5 $D.log; exit 1; }
cat $D.log | grep median
python3 - "$D" <<'PY'
import csv, glob, sys, statistics
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur, gap = {}, []
prev_end, prev_name = None, None
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-40:]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur.setdefault(n, []).append(e - s)
    if prev_end is not None and ("lbfgs_step" in n or "lbfgs_step" in prev_name):
        gap.append(s - prev_end)
    prev_end, prev_name = e, n
for n, v in dur.items():
    print(f"{n:42s} calls {len(v):5d}  median {statistics.median(v)/1e3:7.2f} us  mean {statistics.mean(v)/1e3:7.2f}")
g = [x for x in gap if x < 200000]
print(f"gaps next to a step kernel: n {len(g)}  median {statistics.median(g)/1e3:.2f} us  mean {statistics.mean(g)/1e3:.2f}")
PY
