#!/bin/bash
# tools/libk2b_fstamp.so: the library with per-phase stamps in the split shapes of the fused fit kernel (tools/fit_diag.h).
# usage: tools/build_fit_stamps.sh ; then on the GPU box: python tools/dev_fit_once.py 1024 - 1 fstamp
set -e
cd "$(dirname "$0")/../keypoints2body_amd/csrc"
make -s -j8
DIAG="$(cd ../../tools && pwd)/fit_diag.h"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize \
    -DK2B_FIT_DIAG_HEADER="\"$DIAG\"" -c k2b_fit.hip -o /tmp/k2b_fit_fstamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libk2b_fstamp.so k2b_api.o /tmp/k2b_fit_fstamp.o k2b_fit_tree.o k2b_lbs.o \
    k2b_lbs_stream.o k2b_precompute.o k2b_metrics.o k2b_vertex.o k2b_lbfgs.o
echo "built tools/libk2b_fstamp.so"
