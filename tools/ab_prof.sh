#!/bin/bash
# A/B kernel-duration comparison inside one gpurun call: tools/ab_prof.sh "<ENV_A>" "<ENV_B>" (env assignments or empty)
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
for tag in A B; do
  if [ $tag = A ]; then E="$1"; else E="$2"; fi
  rm -rf /tmp/prof_$tag
  env $E timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_$tag -- python $ROOT/bench.py --no-cpu-baseline --steps 20 > $ROOT/gpurun_out/prof_$tag.log 2>&1
  DB=$(find /tmp/prof_$tag -name "*.db" | head -1)
  python $ROOT/tools/rocpd_summary.py "$DB" $ROOT/gpurun_out/ks_$tag.md > /dev/null
done
for i in 1 2; do
  env $1 timeout 120 python $ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>&1 | grep metric | cut -c1-120 > $ROOT/gpurun_out/bench_A$i.txt
  env $2 timeout 120 python $ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>&1 | grep metric | cut -c1-120 > $ROOT/gpurun_out/bench_B$i.txt
done
