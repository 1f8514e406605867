#!/bin/bash
# tile-shape sweep on the GPU box: bash tools_sweep.sh <workload> "<G:BLOCK pairs>"
for gb in $2; do
  g=${gb%%:*}; b=${gb##*:}
  GMPE_G=$g GMPE_BLOCK=$b timeout -k 10 100 python bench.py --workload $1 --steps 400 --warmup 30 --no-cpu-baseline > gpurun_out/sweep.json 2>gpurun_out/sweep.err
  python -c "import json; d=json.loads(open('gpurun_out/sweep.json').read().strip().splitlines()[-1]); print('$1 G=$g BLOCK=$b', round(d['ms_per_step']*1e3,2), 'us', round(d['roofline']['frac'],4))"
done
