# EdgeTransition z cache policy (E4_ZPOL builds): memory-side traffic per launch, separate PMC passes (no sys/hip tracing)
R=$PWD; OUT=$R/gpurun_out/r5i; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in base zpol2 zpol5 zpol3; do
  if [ $v = base ]; then L=""; else L="FDIPT_LIB=$R/framedipt_amd/lib/libfdipt_hip_$v.so"; fi
  for spec in "FETCH:FETCH_SIZE" "WRITE:WRITE_SIZE"; do
    tag=${spec%%:*}; ctr=${spec#*:}
    rm -rf /tmp/prof_$tag
    env $L rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/prof_$tag -- python $R/bench.py --steps 4 --warmup 1 --eager --no-cpu-baseline --no-reference-precision --no-all-samples > /dev/null 2>&1
    CSV=$(find /tmp/prof_$tag -name "*counter_collection.csv" | head -1)
    [ -n "$CSV" ] && python $R/tools/pmc_summary.py "$CSV" > $OUT/pmc_${v}_$tag.md
  done
done
ls $OUT
