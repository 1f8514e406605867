#!/bin/bash
# round-4 evidence runs, part E (final library): the rows part D did not repeat — rot_inv in three processes, two / three_phase, c4, the c5 shard
set -x
O=gpurun_out/r04e; mkdir -p $O
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --workload c3r --steps 1000 --warmup 50 --no-cpu-baseline --no-boundary > $O/r04_bench_c3r_$i.json 2> $O/c3r_$i.err
done
for w in c3p2 c3p3; do
  timeout -k 10 300 python bench.py --workload $w --steps 1000 --warmup 50 --no-cpu-baseline --no-boundary > $O/r04_bench_$w.json 2> $O/$w.err
done
timeout -k 10 400 python bench.py --workload c4 --steps 100 --warmup 10 --no-boundary > $O/r04_bench_c4.json 2> $O/c4.err
timeout -k 10 400 python bench.py --workload c5 --envs 2048 --steps 100 --warmup 10 --no-boundary > $O/r04_bench_c5.json 2> $O/c5.err
tail -c 200 $O/*.err
