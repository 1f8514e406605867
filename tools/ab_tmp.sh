#!/bin/bash
run() { timeout -k 10 300 python bench.py --workload $2 --envs $3 --steps $4 --warmup 3 --reps 3 --no-cpu-baseline --no-boundary --no-closed-loop > gpurun_out/r2_xs_$1.json 2>gpurun_out/r2_xs.err || { echo fail $1; tail -3 gpurun_out/r2_xs.err; }; }
for cfg in "8 2" "8 3" "6 2" "4 1" "4 2" "4 3" "3 2" "2 1" "5 2"; do set -- $cfg
  GMPE_XSTEP=1 GMPE_CHUNKS=$1 GMPE_AHEAD=$2 run sh_c$1_D$2 c5 2048 40
done
for cfg in "32 2" "16 2" "16 1" "24 2"; do set -- $cfg
  GMPE_XSTEP=1 GMPE_CHUNKS=$1 GMPE_AHEAD=$2 run full_c$1_D$2 c5 16384 10
done
