#!/bin/bash
# A/B of library builds on the GPU box: bash tools/ab.sh "<lib tags>" "<workloads>"   (tag "new" = libgmpe.so, else libgmpe_<tag>.so)
for w in $2; do for t in $1; do
  if [ "$t" = new ]; then lib=contracts-marl-aam-corridors_amd/libgmpe.so; else lib=contracts-marl-aam-corridors_amd/libgmpe_$t.so; fi
  GMPE_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload $w --steps 400 --warmup 40 --no-cpu-baseline > gpurun_out/ab_${t}_$w.json 2>gpurun_out/ab_${t}_$w.err
  python -c "import json; d=json.loads(open('gpurun_out/ab_${t}_$w.json').read().strip().splitlines()[-1]); print('$w', '$t', round(d['ms_per_step']*1e3,2), 'us', round(d['roofline']['frac'],4))"
done; done
