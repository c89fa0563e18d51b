#!/bin/bash
# A/B of two library builds inside one gpurun call: tools/ab2.sh <libA.so> <libB.so> [bench args] -> gpurun_out/ab_{A,B}.md (kernel tables), ab_bench.txt
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
LA=$1; LB=$2; shift 2
cd /tmp && export TMPDIR=/tmp
X="--no-cpu-baseline --no-reference-precision --no-all-samples"
: > $ROOT/gpurun_out/ab_bench.txt
for i in 1 2 3; do
  for tag in A B; do
    if [ $tag = A ]; then L=$LA; else L=$LB; fi
    echo -n "$tag " >> $ROOT/gpurun_out/ab_bench.txt
    FDIPT_LIB=$ROOT/$L python $ROOT/bench.py $X --steps 100 "$@" 2>/dev/null | python $ROOT/tools/print_bench.py >> $ROOT/gpurun_out/ab_bench.txt
  done
done
for tag in A B; do
  if [ $tag = A ]; then L=$LA; else L=$LB; fi
  rm -rf /tmp/prof_$tag
  FDIPT_LIB=$ROOT/$L rocprofv3 --kernel-trace -d /tmp/prof_$tag -- python $ROOT/bench.py $X --steps 40 "$@" > /tmp/prof_$tag.log 2>&1
  DB=$(find /tmp/prof_$tag -name "*.db" | head -1)
  python $ROOT/tools/rocpd_summary.py "$DB" $ROOT/gpurun_out/ab_$tag.md > /dev/null
done
cat $ROOT/gpurun_out/ab_bench.txt
