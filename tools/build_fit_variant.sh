#!/bin/bash
# tools/libk2b_<name>.so: the library with k2b_fit.hip compiled with extra flags (A/B runs in one call: tools/dev_fit_once.py ... <name>).
# usage: tools/build_fit_variant.sh name "-DFLAG=1 ..."
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../keypoints2body_amd/csrc"
make -s -j8
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize \
    $FLAGS -c k2b_fit.hip -o /tmp/k2b_fit_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libk2b_$NAME.so k2b_api.o /tmp/k2b_fit_$NAME.o k2b_fit_tree.o k2b_lbs.o \
    k2b_lbs_stream.o k2b_precompute.o k2b_metrics.o k2b_vertex.o k2b_lbfgs.o
echo "built tools/libk2b_$NAME.so"
