#!/bin/bash
# How often does the two-stream bit-identity test fail with a given library build?  tools/flaky_concurrent.sh <runs> <variant ...>   ("product" = the default library)
R="${GRAFT_REPO_ROOT:-$(pwd)}"; runs=$1; shift
for v in "$@"; do
  if [ "$v" = product ]; then unset FDIPT_LIB; else export FDIPT_LIB="$R/framedipt_amd/lib/libfdipt_hip_$v.so"; fi
  f=0
  for i in $(seq $runs); do
    (cd "$R" && timeout 300 python -m pytest tests/test_gpu_robustness.py -q -x -k "concurrent_forwards or streamed_sub_batches" > /tmp/flaky.log 2>&1) || f=$((f + 1))
  done
  echo "$v: $f of $runs runs failed"
done
