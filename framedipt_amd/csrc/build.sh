#!/bin/bash
# Build libfdipt_hip.so for gfx950 (MI355X).  Cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
# FDIPT_DEV=1: development build (-DFDIPT_DEV: the FDIPT_* environment switches of model.hip / kernels.hpp are read, once per
# process) -> lib/libfdipt_hip_dev.so, loaded instead of the product library when FDIPT_LIB points at it (tools/)
BUILD="$HERE/build"; LIBNAME=libfdipt_hip.so; EXTRA=""
if [ -n "${FDIPT_DEV:-}" ]; then BUILD="$HERE/build_dev"; LIBNAME=libfdipt_hip_dev.so; EXTRA="-DFDIPT_DEV"; fi
# FDIPT_VARIANT=name FDIPT_EXTRA="-D..." : an A/B build with extra compile flags -> lib/libfdipt_hip_<name>.so (loaded through FDIPT_LIB)
if [ -n "${FDIPT_VARIANT:-}" ]; then BUILD="$HERE/build_$FDIPT_VARIANT"; LIBNAME=libfdipt_hip_$FDIPT_VARIANT.so; EXTRA="$EXTRA ${FDIPT_EXTRA:-}"; fi
mkdir -p "$OUT" "$BUILD"
# FDIPT_CLEAN=1 (what __graft_entry__.build() sets): drop every object first, so that the run proves the tree compiles
# (the default is mtime-incremental: objects and libraries travel with the tree to the GPU box)
if [ -n "${FDIPT_CLEAN:-}" ]; then rm -f "$BUILD"/*.o "$OUT/$LIBNAME"; fi
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -Wno-inline-asm: the LDS-DMA helpers name m0 (a reserved register hipcc re-materialises before each of its own uses) as clobbered
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -Wno-inline-asm $EXTRA"
pids=()
T0=$SECONDS
NCOMP=0
for f in gemm ipa_proj2 pair_mlp edge_embed2 edge_transition3 edge_transition4 attention attention3 pair_bias attention_seq chain rowblock frames model; do
  if [ ! -f "$BUILD/$f.o" ] || [ "$HERE/$f.hip" -nt "$BUILD/$f.o" ] || [ "$HERE/common.hpp" -nt "$BUILD/$f.o" ] || [ "$HERE/kernels.hpp" -nt "$BUILD/$f.o" ] || [ "$HERE/../../include/fdipt.h" -nt "$BUILD/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$BUILD/$f.o" &
    pids+=($!)
    NCOMP=$((NCOMP + 1))
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/$LIBNAME" "$BUILD"/{gemm,ipa_proj2,pair_mlp,edge_embed2,edge_transition3,edge_transition4,attention,attention3,pair_bias,attention_seq,chain,rowblock,frames,model}.o
echo "compiled $NCOMP of 14 units for gfx950 in $((SECONDS - T0)) s: $(cd "$BUILD" && ls *.o | tr '\n' ' ')"
echo "built $OUT/$LIBNAME"
