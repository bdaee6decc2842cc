#!/bin/bash
# End-of-state measurement bundle (run on the GPU box): tools/profile_round.sh TAG
# default bench line (4096-frame sequence, strong scaling), the 1024-frame point, SMPL-X, rocprofv3 kernel stats of the
# DEFAULT command and the HBM PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, kernel-trace only beside --pmc).
set -e
TAG=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "default bench done"
python3 $R/bench.py --total-frames 1024 --no-cpu-baseline --no-weak-line > $O/bench_1024.json 2> $O/bench_1024.err
python3 $R/bench.py --model smplx --total-frames 1024 --no-cpu-baseline --no-weak-line > $O/bench_smplx_1024.json 2> $O/bench_smplx_1024.err
python3 $R/bench.py --model smplx --no-cpu-baseline --no-weak-line > $O/bench_smplx_4096.json 2> $O/bench_smplx_4096.err
echo "bench lines done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats4096 -- python3 $R/bench.py --no-cpu-baseline --no-weak-line > $O/bench_default_under_rocprof.json 2> $O/rocprof4096.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1024 -- python3 $R/bench.py --total-frames 1024 --no-cpu-baseline --no-weak-line > $O/bench_1024_under_rocprof.json 2> $O/rocprof1024.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/statsx1024 -- python3 $R/bench.py --model smplx --total-frames 1024 --no-cpu-baseline --no-weak-line > $O/bench_smplx_1024_under_rocprof.json 2> $O/rocprofx1024.err
echo "kernel stats done"
for fr in 4096 1024; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${c}_$fr -- python3 $R/bench.py --steps 5 --warmup 2 --total-frames $fr --no-cpu-baseline --no-weak-line > /dev/null 2> $O/${c}_$fr.err
  done
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${c}_x1024 -- python3 $R/bench.py --model smplx --steps 5 --warmup 2 --total-frames 1024 --no-cpu-baseline --no-weak-line > /dev/null 2> $O/${c}_x1024.err
done
echo "pmc passes done"
bash $R/tools/pmc_fit.sh ${TAG}_4096 4096 > $O/sq_fit_4096.txt 2>&1 || true
bash $R/tools/pmc_fit.sh ${TAG}_1024 1024 > $O/sq_fit_1024.txt 2>&1 || true
echo "sq passes done"
cat $O/bench_default.json
