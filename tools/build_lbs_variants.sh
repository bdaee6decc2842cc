#!/bin/bash
# Timing-only variants of the LBS tile kernel: tools/libk2b_<name>.so (git-ignored; they travel to the GPU box).
# The diagnostics themselves live in tools/lbs_diag.h (hooks the shipped kernel leaves empty).
# APIFLAGS: the definitions also reach k2b_api.hip.
# usage: tools/build_lbs_variants.sh name:"-DFLAGS" ...     e.g.  nostore:"-DK2B_TILE_DIAG=1" chunk4:"-DK2B_TILE_CHUNK=4"
set -e
cd "$(dirname "$0")/../keypoints2body_amd/csrc"
make -s -j8
DIAG="$(cd ../../tools && pwd)/lbs_diag.h"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -DK2B_LBS_DIAG_HEADER=\"$DIAG\""
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  # (-DK2B_LBS_STREAM=0 is a switch of k2b_api.hip: that object is rebuilt with the same definitions)
  /opt/rocm/bin/hipcc $FLAGS $defs -c k2b_api.hip -o /tmp/k2b_api_$name.o
  /opt/rocm/bin/hipcc $FLAGS $defs -c k2b_lbs_stream.hip -o /tmp/k2b_lbs_stream_$name.o
  /opt/rocm/bin/hipcc $FLAGS $defs -c k2b_lbs.hip -o /tmp/k2b_lbs_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libk2b_$name.so /tmp/k2b_api_$name.o k2b_fit.o k2b_fit_tree.o /tmp/k2b_lbs_$name.o /tmp/k2b_lbs_stream_$name.o k2b_precompute.o k2b_metrics.o k2b_vertex.o k2b_lbfgs.o
  echo "built tools/libk2b_$name.so ($defs)"
done
