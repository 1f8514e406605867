#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline line is checked against, for ONE workload, on the GPU box:
#   gpurun_out/prof_<wl>/  --kernel-trace --stats      (average k_env duration)
#   gpurun_out/pmcW_<wl>/  --pmc WRITE_SIZE            (separate passes, MI355X_MICROARCH.md HBM section)
#   gpurun_out/pmcF_<wl>/  --pmc FETCH_SIZE
# usage (inside gpurun): bash tools/profile.sh c2 [steps] ["extra bench flags"]   then, in the container: python tools/profile.py c2 r02 <steps>
# The profiled command is the default bench.py line minus its side measurements (closed-loop / NumPy-boundary / CPU legs), so every
# launch of the dominant kernel in the trace belongs to the timed shape.
set -e
WL=${1:-c2}; STEPS=${2:-300}; EXTRA=${3:-}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for pass in prof pmcW pmcF; do
  case $pass in
    prof) FLAGS="--kernel-trace --stats" ;;
    pmcW) FLAGS="--pmc WRITE_SIZE" ;;
    pmcF) FLAGS="--pmc FETCH_SIZE" ;;
  esac
  rm -rf "$ROOT/gpurun_out/${pass}_${WL}"
  timeout -k 10 300 rocprofv3 $FLAGS -d "$ROOT/gpurun_out/${pass}_${WL}" -o run --output-format csv -- \
      python3 "$ROOT/bench.py" --workload "$WL" --steps "$STEPS" --warmup 20 --reps 3 --no-cpu-baseline --no-boundary --no-closed-loop $EXTRA > "$ROOT/gpurun_out/${pass}_${WL}.log" 2>&1
  echo "$pass $WL done"
done
