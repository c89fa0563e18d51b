#!/bin/bash
# kernel-trace table of a short bench run on the GPU box: tools/kstats.sh <out.md> [bench args...]
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-reference-precision --no-all-samples --steps 40 "$@" > /tmp/prof_kt.log 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py "$DB" $GRAFT_REPO_ROOT/$OUT > /dev/null
