#!/bin/bash
# SQ stall / activity counters per kernel of a short bench run (two --pmc passes with --kernel-trace only): tools/pmc_sq.sh <out.md> [bench args]
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
: > $GRAFT_REPO_ROOT/$OUT
for ctr in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/prof_sq
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/prof_sq -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-precision --no-all-samples "$@" > /tmp/prof_sq.log 2>&1
  CSV=$(find /tmp/prof_sq -name "*counter_collection.csv" | head -1)
  if [ -n "$CSV" ]; then python $GRAFT_REPO_ROOT/tools/pmc_table.py "$CSV" >> $GRAFT_REPO_ROOT/$OUT; else echo "no csv for: $ctr" >> $GRAFT_REPO_ROOT/$OUT; tail -5 /tmp/prof_sq.log >> $GRAFT_REPO_ROOT/$OUT; fi
done
