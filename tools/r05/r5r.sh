mkdir -p gpurun_out/r5r
F="--no-cpu-baseline --no-reference-precision --no-all-samples"
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b64 && rocprofv3 --kernel-trace --stats -d /tmp/prof_b64 -- python $R/bench.py --samples-per-gpu 64 --steps 18 --warmup 2 $F > $R/gpurun_out/r5r/b64.log 2>&1
DB=$(find /tmp/prof_b64 -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_summary.py "$DB" $R/gpurun_out/r5r/kernel_stats_b64.md > /dev/null
tail -2 $R/gpurun_out/r5r/b64.log | head -c 300
