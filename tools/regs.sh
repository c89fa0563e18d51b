#!/bin/bash
# compile one translation unit of the library stand-alone and list registers / spills per kernel (developer helper)
#   tools/regs.sh rowblock [extra hipcc flags]
f=$1; shift
mkdir -p /tmp/t
cd /root/repo/framedipt_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-inline-asm "$@" -c $f.hip -o /tmp/t/$f.o -save-temps=obj 2>&1 | grep -v "^$" | head -30
cd /tmp/t && grep -E "\.name:|vgpr_count|vgpr_spill|sgpr_spill" $f-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - | awk '{print $2, "sgpr_spill", $4, "vgpr", $6, "vgpr_spill", $8}'
