#!/bin/bash
# round-4 evidence runs, part D (final library: batched placement in the navigation_graph / July rollout kernels): full GPU suite + smoke, the driver's command and the
# long-K lines of c2 / c3 in three separate processes each, K sweeps, rocprofv3 stats + HBM PMC of c2 / c3, SQ counters of c2, nontemporal-vs-plain rollout stores for rot_inv in one process
set -x
O=gpurun_out/r04d; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 $([ $i -gt 1 ] && echo --no-cpu-baseline --no-boundary) > $O/r04_default_run_bench_line_$i.json 2> $O/default_$i.err
done
for w in c2 c3; do for i in 1 2 3; do
  timeout -k 10 300 python bench.py --workload $w --steps 1000 --warmup 50 --no-cpu-baseline --no-boundary > $O/r04_bench_${w}_$i.json 2> $O/${w}_$i.err
done; done
timeout -k 10 200 python tools/ksweep.py c2 $O/r04_c2_vs_K.json
timeout -k 10 200 python tools/ksweep.py c3 $O/r04_c3_vs_K.json
timeout -k 10 200 python tools/abinproc.py c3r GMPE_ROLLNT 0 1 300 > $O/rollnt_c3r.log 2>&1
bash tools/profile.sh c2 300
bash tools/profile.sh c3 300
bash tools/pmc_sq.sh c2 300
tail -c 200 $O/*.err
