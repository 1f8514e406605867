#!/bin/bash
# round-4 evidence runs, part B: rocprofv3 kernel stats + HBM PMC of the default bench launch (c2, c3), the all-16384-env c5 line, 2-rank rehearsals of the multi-GPU path
set -x
O=gpurun_out/r04; mkdir -p $O
bash tools/profile.sh c2 300
bash tools/profile.sh c3 300
timeout -k 10 500 python bench.py --workload c5 --steps 30 --warmup 5 --no-boundary --no-cpu-baseline > $O/r04_bench_c5_16384.json 2> $O/c5full.err
for w in c2 c3 c3r; do
  GMPE_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
      bench.py --gpus 2 --steps 100 --warmup 10 --workload $w --gather --no-cpu-baseline > $O/r04_rehearsal_$w.json 2> $O/reh_$w.err
done
tail -c 300 $O/*.err
