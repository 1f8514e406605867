#!/bin/bash
# round-3 evidence runs (one gpurun call): K sweeps, the driver's default command, long-K lines per workload, 2-rank rehearsals
set -x
mkdir -p gpurun_out/r03
timeout -k 10 200 python tools/ksweep.py c2 gpurun_out/r03/r03_c2_vs_K.json
timeout -k 10 200 python tools/ksweep.py c3 gpurun_out/r03/r03_c3_vs_K.json
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/r03_default_run_bench_line.json 2> gpurun_out/r03/default.err
for w in c2 c3 c3r c3p2 c3p3; do
  timeout -k 10 300 python bench.py --workload $w --steps 1000 --warmup 50 > gpurun_out/r03/r03_bench_$w.json 2> gpurun_out/r03/$w.err
done
timeout -k 10 400 python bench.py --workload c4 --steps 100 --warmup 10 > gpurun_out/r03/r03_bench_c4.json 2> gpurun_out/r03/c4.err
timeout -k 10 400 python bench.py --workload c5 --envs 2048 --steps 100 --warmup 10 > gpurun_out/r03/r03_bench_c5.json 2> gpurun_out/r03/c5.err
timeout -k 10 500 python bench.py --workload c5 --steps 30 --warmup 5 --no-boundary > gpurun_out/r03/r03_bench_c5_16384.json 2> gpurun_out/r03/c5full.err
for w in c3 c3r; do
  GMPE_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
      bench.py --gpus 2 --steps 100 --warmup 10 --workload $w --gather --no-cpu-baseline > gpurun_out/r03/r03_rehearsal_$w.json 2> gpurun_out/r03/reh_$w.err
done
tail -c 300 gpurun_out/r03/*.err
