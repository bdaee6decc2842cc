#!/bin/bash
# time the LBS forward with several library variants, each in its own process: tools/dev_lbs_variants.sh FRAMES[:smplx] name...
FR=${1%%:*}; KIND=smpl; [[ "$1" == *:* ]] && KIND=${1#*:}; shift
for n in "$@"; do timeout -k 10 120 python3 $GRAFT_REPO_ROOT/tools/dev_lbs_time.py $n $FR $KIND 2>&1 | grep frames; done
