#!/bin/bash
# Build libfdipt_hip.so for gfx950 (MI355X).  Cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/build"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -Wno-inline-asm: the LDS-DMA helpers name m0 (a reserved register hipcc re-materialises before each of its own uses) as clobbered
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -Wno-inline-asm"
pids=()
for f in gemm ipa_proj2 pair_mlp edge_embed2 edge_transition3 edge_transition4 attention attention3 pair_bias attention_seq chain rowblock frames model; do
  if [ ! -f "$HERE/build/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/build/$f.o" ] || [ "$HERE/common.hpp" -nt "$HERE/build/$f.o" ] || [ "$HERE/kernels.hpp" -nt "$HERE/build/$f.o" ] || [ "$HERE/../../include/fdipt.h" -nt "$HERE/build/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/build/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libfdipt_hip.so" "$HERE"/build/{gemm,ipa_proj2,pair_mlp,edge_embed2,edge_transition3,edge_transition4,attention,attention3,pair_bias,attention_seq,chain,rowblock,frames,model}.o
echo "built $OUT/libfdipt_hip.so"
