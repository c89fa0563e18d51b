#!/bin/bash
# Sample socket power and shader clock (rocm-smi) while a command runs: tools/power_sample.sh <out file> <command ...>
out=$1; shift
"$@" > /tmp/ps_cmd.out 2>&1 &
pid=$!
sleep ${PS_DELAY:-25}
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | tr '\n' ' ' >> "$out"; echo >> "$out"
  sleep 0.3
done
wait $pid
tail -3 /tmp/ps_cmd.out
