#!/bin/bash
# round 4: placement of the next episode ahead of the reset (GMPE_PRE) — parity first, then A/B against a build without the code (libgmpe_nopre.so) and through the knob
set -o pipefail
python -m pytest tests -m gpu -x -q -k "rollout or random_config or instantiations or gather" > gpurun_out/r4_pre_tests.log 2>&1 || { tail -30 gpurun_out/r4_pre_tests.log; exit 1; }
tail -3 gpurun_out/r4_pre_tests.log
bash tools/abk.sh "nopre new nopre new" "c2 c3" "20,300" 2>&1 | tee gpurun_out/r4_pre_abk.log
for w in c2 c3 c3r; do for K in 20 300; do timeout -k 10 200 python tools/abinproc.py $w GMPE_PRE 0 1 $K 2>&1 | grep -v amdgpu.ids; done; done | tee gpurun_out/r4_pre_abinproc.log
