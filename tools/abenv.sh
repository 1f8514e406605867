#!/bin/bash
# A/B of an environment knob through the K sweep, alternating inside one call: bash tools/abenv.sh GMPE_FUSE "0 1 0 1" "c2" "20,300"
for w in $3; do for v in $2; do
  echo "== $w $1=$v"
  env $1=$v KSWEEP_K=${4:-20,300} timeout -k 10 150 python tools/ksweep.py $w 2>&1 | grep -v amdgpu.ids
done; done
