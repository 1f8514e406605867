#!/bin/bash
# round-4 evidence, part C: rocprofv3 kernel stats + HBM PMC of the default bench launch with the FINAL library (compile-time-G kernels), SQ counters of the c2 rollout
set -x
bash tools/profile.sh c2 300
bash tools/profile.sh c3 300
bash tools/pmc_sq.sh c2 300
