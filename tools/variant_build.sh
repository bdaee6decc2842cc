#!/bin/bash
# Diagnostic builds of libk2b with extra -D flags: tools/variant_build.sh <name> <flags...>
set -e
name=$1; shift
cd "$(dirname "$0")/../keypoints2body_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize $*"
mkdir -p /tmp/k2b_$name
for f in k2b_api k2b_fit k2b_lbs k2b_precompute; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o /tmp/k2b_$name/$f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libk2b_$name.so /tmp/k2b_$name/*.o
echo built tools/libk2b_$name.so
