#!/bin/bash
# Collect the round's judged profiles on the GPU box (run through gpurun from the repo root):
#   gpurun_out/final/{bench*.json, kernel_stats.md, pmc_FETCH.md, pmc_WRITE.md, pmc_MFMA.md}
# PMC passes are separate --pmc runs with --kernel-trace only (never combined with sys/hip tracing), as MI355X_MICROARCH.md
# prescribes.  The kernel-stats run is the SAME command as the bench (whole trajectory), so that its EdgeTransition average agrees
# with the HIP-event timing on the bench's JSON line.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/final"
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench.err"   # the driver's exact command
python "$ROOT/bench.py" > "$OUT/bench.json" 2>> "$OUT/bench.err"
for c in c2 c3 c3e c3w c5; do python "$ROOT/bench.py" --config $c --no-cpu-baseline > "$OUT/bench_$c.json" 2>> "$OUT/bench.err"; done
python "$ROOT/bench.py" --precision fp32 --steps 6 --warmup 2 --no-cpu-baseline --no-reference-precision > "$OUT/bench_c4_fp32.json" 2>> "$OUT/bench.err"
python "$ROOT/bench.py" --config c5 --precision fp16 --no-cpu-baseline > "$OUT/bench_c5_fp16.json" 2>> "$OUT/bench.err"
python "$ROOT/bench.py" --kernel-flags 32 --no-cpu-baseline > "$OUT/bench_c4_nosplit.json" 2>> "$OUT/bench.err"
python "$ROOT/bench.py" --samples-per-gpu 24 --steps 100 --no-cpu-baseline --no-reference-precision > "$OUT/bench_c4_b24.json" 2>> "$OUT/bench.err"
python "$ROOT/bench.py" --samples-per-gpu 64 --steps 40 --no-cpu-baseline --no-reference-precision > "$OUT/bench_c4_b64.json" 2>> "$OUT/bench.err"
python "$ROOT/bench.py" --eager --no-cpu-baseline --no-reference-precision --no-all-samples > "$OUT/bench_c4_eager.json" 2>> "$OUT/bench.err"
rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python "$ROOT/bench.py" --no-cpu-baseline --no-reference-precision --no-all-samples > "$OUT/bench_prof.log" 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
[ -n "$DB" ] && python "$ROOT/tools/rocpd_summary.py" "$DB" "$OUT/kernel_stats.md" > /dev/null
# all 64 samples of configs[3] on the one GPU: the kernel table behind `all_samples_one_gpu`
rm -rf /tmp/prof_b64 && rocprofv3 --kernel-trace --stats -d /tmp/prof_b64 -- python "$ROOT/bench.py" --samples-per-gpu 64 --steps 18 --warmup 2 --no-cpu-baseline --no-reference-precision --no-all-samples > "$OUT/bench_b64_prof.log" 2>&1
DB=$(find /tmp/prof_b64 -name "*.db" | head -1)
[ -n "$DB" ] && python "$ROOT/tools/rocpd_summary.py" "$DB" "$OUT/kernel_stats_b64.md" > /dev/null
# config 5 (fp32 mode, N = 1000): kernel table of the parity mode's heaviest configuration
rm -rf /tmp/prof_c5 && rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -- python "$ROOT/bench.py" --config c5 --no-cpu-baseline > "$OUT/bench_c5_prof.log" 2>&1
DB=$(find /tmp/prof_c5 -name "*.db" | head -1)
[ -n "$DB" ] && python "$ROOT/tools/rocpd_summary.py" "$DB" "$OUT/kernel_stats_c5.md" > /dev/null
for spec in "FETCH:FETCH_SIZE" "WRITE:WRITE_SIZE" "MFMA:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=${spec%%:*}; ctr=${spec#*:}
  rm -rf /tmp/prof_$tag
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/prof_$tag -- python "$ROOT/bench.py" --steps 4 --warmup 1 --eager --no-cpu-baseline --no-reference-precision --no-all-samples > /dev/null 2>&1
  CSV=$(find /tmp/prof_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$CSV" ] && python "$ROOT/tools/pmc_summary.py" "$CSV" > "$OUT/pmc_$tag.md"
done
ls -la "$OUT"
