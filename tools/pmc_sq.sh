#!/bin/bash
# SQ issue / stall counters of the dominant kernel (one pass, 8 SQ slots): bash tools/pmc_sq.sh c2 [steps] ["extra bench flags"]
set -e
WL=${1:-c2}; STEPS=${2:-300}; EXTRA=${3:-}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for pass in sqA sqB; do
  case $pass in
    sqA) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" ;;
    sqB) C="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" ;;
  esac
  rm -rf "$ROOT/gpurun_out/${pass}_${WL}"
  timeout -k 10 300 rocprofv3 --pmc $C -d "$ROOT/gpurun_out/${pass}_${WL}" -o run --output-format csv -- \
      python3 "$ROOT/bench.py" --workload "$WL" --steps "$STEPS" --warmup 20 --reps 3 --no-cpu-baseline --no-boundary --no-closed-loop $EXTRA > "$ROOT/gpurun_out/${pass}_${WL}.log" 2>&1
  echo "$pass $WL done"
done
