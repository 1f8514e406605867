#!/bin/bash
# A/B of library builds through the K sweep (slots26 / one_slot / closed_loop): bash tools/abk.sh "<lib tags>" "<workloads>" "<K list>"
# tag "new" = libgmpe.so, else libgmpe_<tag>.so; builds alternate inside ONE gpurun call (box-to-box spread is larger than most effects)
for w in $2; do for t in $1; do
  if [ "$t" = new ]; then lib=contracts-marl-aam-corridors_amd/libgmpe.so; else lib=contracts-marl-aam-corridors_amd/libgmpe_$t.so; fi
  echo "== $w $t"
  GMPE_LIB=$PWD/$lib KSWEEP_K=${3:-20,300} timeout -k 10 150 python tools/ksweep.py $w gpurun_out/abk_${t}_$w.json 2>&1 | grep -v amdgpu.ids
done; done
