#!/bin/bash
# tools/sweep_env.sh VAR v1 v2 ...: bench line + per-kernel table for each value of an environment variable, inside one gpurun call
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
VAR=$1; shift
cd /tmp && export TMPDIR=/tmp
: > $ROOT/gpurun_out/sweep.txt
for v in "$@"; do
  rm -rf /tmp/prof_s
  env $VAR=$v timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_s -- python $ROOT/bench.py --no-cpu-baseline --steps 20 > /dev/null 2>&1
  DB=$(find /tmp/prof_s -name "*.db" | head -1)
  echo "== $VAR=$v" >> $ROOT/gpurun_out/sweep.txt
  python $ROOT/tools/rocpd_summary.py "$DB" | grep "${SWEEP_GREP:-splitk\|layernorm}" | cut -c1-120 >> $ROOT/gpurun_out/sweep.txt
  env $VAR=$v timeout 120 python $ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>&1 | grep metric | cut -c40-100 >> $ROOT/gpurun_out/sweep.txt
done
