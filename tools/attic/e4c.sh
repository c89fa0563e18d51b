#!/bin/bash
# compile edge_transition4.hip stand-alone and summarise registers / spills (developer helper)
cd /root/repo/framedipt_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-inline-asm "$@" -c edge_transition4.hip -o /tmp/t/et4.o -save-temps=obj 2>&1 | grep -v "^$" | head -20
cd /tmp/t; S=edge_transition4-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "\.name:|vgpr_count|vgpr_spill" $S | paste - - - | tail -1
awk '/^_Z23edge_transition4_kernel/,/s_endpgm/' $S > e4.s; wc -l e4.s
grep -n "s_barrier\|scratch_store\|scratch_load" e4.s | awk '{print $1,$2}' | tr '\n' ' ' | fold -w 200 | head -40
