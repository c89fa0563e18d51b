#!/bin/bash
# Per-kernel average durations of the bench step under rocprofv3 (run through gpurun from the repo root):
#   tools/kernel_times.sh <tag> [bench.py arguments ...]      -> gpurun_out/kt/<tag>.md ; environment variables pass through
tag=$1; shift
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p "$ROOT/gpurun_out/kt"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$tag
rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -- python "$ROOT/bench.py" --no-cpu-baseline --no-all-samples --steps 20 --warmup 2 "$@" > /tmp/kt_$tag.log 2>&1
DB=$(find /tmp/kt_$tag -name "*.db" | head -1)
python "$ROOT/tools/rocpd_summary.py" "$DB" "$ROOT/gpurun_out/kt/$tag.md" > /dev/null
