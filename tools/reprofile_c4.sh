set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$ROOT/gpurun_out/final"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python "$ROOT/bench.py" --no-cpu-baseline --no-reference-precision --no-all-samples > "$OUT/bench_prof.log" 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1); [ -n "$DB" ] && python "$ROOT/tools/rocpd_summary.py" "$DB" "$OUT/kernel_stats.md" > /dev/null
for spec in "FETCH:FETCH_SIZE" "WRITE:WRITE_SIZE" "MFMA:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=${spec%%:*}; ctr=${spec#*:}; rm -rf /tmp/prof_$tag
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/prof_$tag -- python "$ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-reference-precision --no-all-samples > /dev/null 2>&1
  CSV=$(find /tmp/prof_$tag -name "*counter_collection.csv" | head -1); [ -n "$CSV" ] && python "$ROOT/tools/pmc_summary.py" "$CSV" > "$OUT/pmc_$tag.md"
done
python "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
