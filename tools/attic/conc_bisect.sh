#!/bin/bash
# bisect the concurrent-forward mismatch over the dev switches (FDIPT_DEV build); run through gpurun from the repo root:
#   tools/conc_bisect.sh "FDIPT_ATTN_V1=1 FDIPT_NO_SEQ_ATTN=1" "FDIPT_ET_V1=1" ...      (one quoted group of assignments per run)
export FDIPT_LIB=$PWD/framedipt_amd/lib/libfdipt_hip_dev.so
run() { tag="$1"; out=$(env $1 python tools/conc_forward_check.py ${CONC_N:-300} 8 ${CONC_REPS:-80} 2>/dev/null); n=$(echo "$out" | grep -c "rows differing"); echo "$tag: $n of $((2 * ${CONC_REPS:-80})) half-forwards differ"; }
for sw in "$@"; do run "$sw"; done
