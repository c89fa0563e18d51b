#!/bin/bash
# One parametrised runner for everything a gpurun call does on the GPU box (round 6: replaces the one-off tools/r05/r5*.sh scripts).
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh <tag> <step> [<step> ...]'      -> gpurun_out/<tag>/
# steps (run in the order given; FDIPT_* / other environment variables pass through to every step):
#   tests[:<pytest -k expression>]   pytest -m gpu -x -q                         -> tests.log
#   smoke                            __graft_entry__.smoke()                      -> smoke.log
#   bench[:<extra bench.py args>]    the driver's command (--steps 20 --warmup 5) -> bench.json (one JSON line) + bench.log
#   fast[:<extra args>]              bench without the sub-records, 60 steps       -> fast.txt (appended: repeat the step for A/B/A/B)
#   kstats[:<extra args>]            rocprofv3 --kernel-trace --stats, 8 samples   -> kernel_stats.md
#   kstats64                         ... 64 samples on the GPU                     -> kernel_stats_b64.md
#   kstats32                         ... fp32 mode (reference arithmetic)          -> kernel_stats_fp32.md
#   pmc:<counters>                   rocprofv3 --pmc <counters> (own pass, kernel-trace only) -> pmc_<first counter>.md
#   py:<script and args>             python <script ...>                          -> py_<n>.log
#   lib:<variant>                    export FDIPT_LIB=framedipt_amd/lib/libfdipt_hip_<variant>.so for the following steps (lib: = product library)
tag=$1; shift
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/$tag"
mkdir -p "$O"
export TMPDIR=/tmp
F="--no-cpu-baseline --no-reference-precision --no-all-samples"
npy=0
prof() {  # prof <out.md> <bench args...>
  local out=$1; shift
  rm -rf /tmp/prof_kt
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python "$R/bench.py" $F "$@" > "$O/prof.log" 2>&1)
  local db; db=$(find /tmp/prof_kt -name "*.db" | head -1)
  [ -n "$db" ] && python "$R/tools/rocpd_summary.py" "$db" "$out" > /dev/null
}
for step in "$@"; do
  name=${step%%:*}; arg=""; [ "$name" != "$step" ] && arg=${step#*:}
  case $name in
    tests) if [ -n "$arg" ]; then (cd "$R" && timeout 1500 python -m pytest tests -m gpu -x -q -k "$arg" > "$O/tests.log" 2>&1); else (cd "$R" && timeout 1500 python -m pytest tests -m gpu -q > "$O/tests.log" 2>&1); fi; tail -3 "$O/tests.log" ;;
    smoke) (cd "$R" && timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1); tail -1 "$O/smoke.log" ;;
    bench) (cd "$R" && timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 $arg > "$O/bench.log" 2>&1); grep '"metric"' "$O/bench.log" | tail -1 > "$O/bench.json"; cut -c1-400 "$O/bench.json" ;;
    fast) (cd "$R" && timeout 300 python bench.py --steps 60 --warmup 5 $F $arg 2>&1 | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('${FDIPT_LIB:-product}', '$arg', d['value'], d['ms_per_step'], d['roofline']['ms'] if 'ms' in d['roofline'] else d['roofline'].get('achieved'))" >> "$O/fast.txt"); tail -1 "$O/fast.txt" ;;
    kstats) prof "$O/kernel_stats.md" --steps 40 $arg ;;
    kstats64) prof "$O/kernel_stats_b64.md" --samples-per-gpu 64 --steps 18 --warmup 2 ;;
    kstats32) prof "$O/kernel_stats_fp32.md" --precision fp32 --steps 12 --warmup 2 ;;
    pmc) rm -rf /tmp/prof_pmc; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $arg -d /tmp/prof_pmc -- python "$R/bench.py" $F --steps 8 --warmup 2 > "$O/pmc.log" 2>&1)
         db=$(find /tmp/prof_pmc -name "*.db" | head -1); [ -n "$db" ] && python "$R/tools/pmc_table.py" "$db" "$O/pmc_${arg%% *}.md" > /dev/null ;;
    py) npy=$((npy + 1)); (cd "$R" && timeout 1200 python $arg > "$O/py_$npy.log" 2>&1); tail -5 "$O/py_$npy.log" ;;
    lib) if [ -n "$arg" ]; then export FDIPT_LIB="$R/framedipt_amd/lib/libfdipt_hip_$arg.so"; else unset FDIPT_LIB; fi ;;
    *) echo "unknown step $step" ;;
  esac
done
