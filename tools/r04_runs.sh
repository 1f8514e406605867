#!/bin/bash
# round-4 evidence runs, part A (one gpurun call): the driver's command in three separate processes, long-K lines in three separate processes per small-E workload
# (README / DESIGN quote the MEDIAN process with the range: VERDICT r3 item 6), c4 / c5 lines, K sweeps.
set -x
O=gpurun_out/r04; mkdir -p $O
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $([ $i -gt 1 ] && echo --no-cpu-baseline --no-boundary) > $O/r04_default_run_bench_line_$i.json 2> $O/default_$i.err
done
for w in c2 c3 c3r; do for i in 1 2 3; do
  timeout -k 10 300 python bench.py --workload $w --steps 1000 --warmup 50 --no-cpu-baseline --no-boundary > $O/r04_bench_${w}_$i.json 2> $O/${w}_$i.err
done; done
for w in c3p2 c3p3; do
  timeout -k 10 300 python bench.py --workload $w --steps 1000 --warmup 50 --no-cpu-baseline --no-boundary > $O/r04_bench_$w.json 2> $O/$w.err
done
timeout -k 10 400 python bench.py --workload c4 --steps 100 --warmup 10 --no-boundary > $O/r04_bench_c4.json 2> $O/c4.err
timeout -k 10 400 python bench.py --workload c5 --envs 2048 --steps 100 --warmup 10 --no-boundary > $O/r04_bench_c5.json 2> $O/c5.err
timeout -k 10 200 python tools/ksweep.py c2 $O/r04_c2_vs_K.json
timeout -k 10 200 python tools/ksweep.py c3 $O/r04_c3_vs_K.json
tail -c 200 $O/*.err
