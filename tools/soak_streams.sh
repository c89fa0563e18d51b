#!/bin/bash
# Two sub-batch streams against the single-stream trajectory, over and over, at the four benchmarked sizes (the concurrency
# that exposed the wide-store hazard of round 2): prints one line per run and a summary.  tools/soak_streams.sh <seconds>
END=$(( $(date +%s) + ${1:-480} ))
ok=0; bad=0
while [ $(date +%s) -lt $END ]; do
  for cfg in "300 8 120" "724 5 30" "1000 4 16" "128 8 200"; do
    if python $GRAFT_REPO_ROOT/tools/streams_long_check.py $cfg > /tmp/soak.log 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); echo "MISMATCH $cfg"; tail -3 /tmp/soak.log; fi
    [ $(date +%s) -lt $END ] || break
  done
done
echo "soak: $ok runs bit-identical (1 / 2 streams and repetition), $bad with a mismatch"
