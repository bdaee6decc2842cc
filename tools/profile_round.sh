#!/bin/bash
# End-of-state measurement bundle (run on the GPU box): tools/profile_round.sh TAG
set -e
TAG=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 30 --warmup 5 > $O/bench_1024.json 2> $O/bench_1024.err
python3 $R/bench.py --steps 30 --warmup 5 --frames 4096 --no-cpu-baseline > $O/bench_4096.json 2> $O/bench_4096.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1024 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_1024_under_rocprof.json 2> $O/rocprof1024.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats4096 -- python3 $R/bench.py --steps 20 --warmup 3 --frames 4096 --no-cpu-baseline > $O/bench_4096_under_rocprof.json 2> $O/rocprof4096.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch1024 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write1024 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/write.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch4096 -- python3 $R/bench.py --steps 5 --warmup 2 --frames 4096 --no-cpu-baseline > /dev/null 2> $O/fetch4096.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write4096 -- python3 $R/bench.py --steps 5 --warmup 2 --frames 4096 --no-cpu-baseline > /dev/null 2> $O/write4096.err
cat $O/bench_1024.json $O/bench_4096.json
