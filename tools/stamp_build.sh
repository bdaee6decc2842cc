#!/bin/bash
# Diagnostic build: libk2b_stamps.so = libk2b with s_memtime stamps in the fit kernel.
set -e
cd "$(dirname "$0")/../keypoints2body_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -DK2B_FIT_STAMPS"
mkdir -p /tmp/k2b_stamps
for f in k2b_api k2b_fit k2b_lbs k2b_precompute; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o /tmp/k2b_stamps/$f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libk2b_stamps.so /tmp/k2b_stamps/*.o
echo built tools/libk2b_stamps.so
