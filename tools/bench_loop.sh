#!/bin/bash
# the bench's workload a few times in a row (for tools/power_sample.sh)
for i in $(seq ${1:-4}); do python "$(dirname "$0")/../bench.py" --no-cpu-baseline | cut -c1-120; done
