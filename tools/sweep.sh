#!/bin/bash
# tile-shape sweep on the GPU box: bash tools/sweep.sh <workload> "<G:BLOCK pairs, or 'auto'>" [envs]
for gb in $2; do
  if [ "$gb" = auto ]; then unset GMPE_G GMPE_BLOCK; g=auto; b=auto; else g=${gb%%:*}; b=${gb##*:}; export GMPE_G=$g GMPE_BLOCK=$b; fi
  timeout -k 10 100 python bench.py --workload $1 ${3:+--envs $3} --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/sweep.json 2>gpurun_out/sweep.err
  python -c "import json; d=json.loads(open('gpurun_out/sweep.json').read().strip().splitlines()[-1]); print('$1 N=${3:-default} G=$g BLOCK=$b', round(d['ms_per_step']*1e3,2), 'us', round(d['roofline']['frac'],4), '%.3e' % d['value'])"
done
